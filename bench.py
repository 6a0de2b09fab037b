#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Atom W4A4 group-128 GEMM on MI355X.

One "step" = one launch of atom_gemm_w4a4_f16 (through the C ABI) on the headline workload of BASELINE.json
(configs[2]: M=N=K=4096, group 128, 128 INT8 outlier columns), operands resident in HBM, synthetic uniform random
int4/int8 codes and U(0.005,0.05) fp16 scales (never zeros: zero data clocks ~19 % higher).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--M 4096 --N 4096 --K 4096] [--format f6|packed|wide] [--no-cpu-baseline]
                    [--no-configs] [--no-block] [--no-decode] [--workload gemm|block]
                    [--ramp 1500]   (untimed set-up launches ahead of the W warm-up steps: power-state ramp after idle;
                                     reported as "ramp" in the JSON line)

--gpus N > 1 without a torch.distributed environment (no WORLD_SIZE): bench.py starts the N ranks itself
(python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>); under
torch.distributed.run WORLD_SIZE must equal --gpus.  "n_gpus" in the JSON line is the number of ranks that ran.

The headline runs the operand format the drop-in modules use for prefill batches (F6: both operands BF6-coded, block-scaled
MFMA); at N=1 the reference's packed-nibble format and the wide-activation format are timed beside it (same codes, same
scales, outputs compared bit for bit) and reported under "other_operand_formats", as is the packed format on the re-coding
route three ways: "packed_ws" (C ABI + caller-owned workspace, nothing cached), "packed_ws_same_weight" (the caller asserts
ATOM_WS_WEIGHT_CACHED: one weight repeated) and "packed_ops" (atom_amd.ops with two weights in alternation: its per-weight
cache; = "abi_value").  Also at N=1: "cold" (the headline with 512 MB flushed between launches) and "configs" -- the other
BASELINE configs (config 2 M=1, the corners of config 5, the 4096-wide batch sweep), HIP-graph replay per launch, hot and with
the weights streamed from HBM (--no-configs skips it), and "block" -- BASELINE config 4 (one Llama-7B decoder block at batch 32 x
seq 2048, KV INT4) through --workload block in its own process, with its GEMM / quantiser / attention split (--no-block skips it).

N > 1: one process per GPU (torch.distributed.run), every rank runs an independent replica (the path is a
single-device per-layer GEMM: "replicas only", no collective on the data path); value = all ranks' ops / max time.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_I8_TOPS = 5033.0      # dense INT8 MFMA: 256 CU x 4 SIMD x 2048 op/clk x 2.4 GHz (MI355X_MICROARCH.md)


def make_operands(M, N, K, device, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    K4 = K - 128
    G = K4 // 128
    u8 = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, generator=g)
    a4, b4 = u8(M, K4 // 2), u8(N, K4 // 2)                      # two uniform int4 codes per byte
    a8, b8 = u8(M, 128).view(torch.int8), u8(N, 128).view(torch.int8)
    sc = lambda *s: (torch.rand(*s, generator=g) * 0.045 + 0.005).half()
    sA, sA8, sB8 = sc(G, M), sc(M), sc(N)
    sB = sc(G, N // 2).repeat_interleave(2, dim=1).contiguous()  # channel pairs share a scale
    return [t.to(device) for t in (a4, b4, sA, sB, a8, b8, sA8, sB8)]


def algorithmic_bytes(M, N, K):
    K4 = K - 128
    G = K4 // 128
    return M * K4 // 2 + N * K4 // 2 + 128 * (M + N) + 2 * (M * G + N * G + M + N) + 2 * M * N


def _median_time(fn, budget_s, max_runs=10):
    t0 = time.perf_counter()
    fn()                                                         # warm-up
    first = time.perf_counter() - t0
    times = []
    start = time.perf_counter()
    while len(times) < max_runs and (time.perf_counter() - start) < budget_s:
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return (float(np.median(times)) if times else first), max(len(times), 1)


def cpu_baseline(M, N, K, budget_s=12.0):
    """The reference's CPU path for this GEMM: QLinearLayer.forward == F.linear on fake-quant FP16
    operands (model/qLinearLayer.py:32-35), torch CPU kernels, all host threads -- the faithful FP16 line is `value`; beside
    it the FP32-upcast line ("best CPU") and the CPU time of the activation quantiser that feeds the GEMM
    (quantize_activation_wrapper, model/quant.py:187-231, restated in oracle.reorder_quant).  Bounded samples."""
    from oracle import atom_oracle as O
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(M, K, generator=g) * 0.5).half()
    w = (torch.randn(N, K, generator=g) * 0.05).half()
    ops = 2.0 * M * N * K
    t16, n16 = _median_time(lambda: O.sim_linear_torch(x, w), budget_s)
    x32, w32 = x.float(), w.float()
    t32, n32 = _median_time(lambda: O.sim_linear_torch(x32, w32), budget_s / 3)
    Mq = min(M, 1024)                                            # quantiser sample: 1024 tokens of the same hidden size
    xq = x[:Mq].numpy()
    idx = np.random.default_rng(0).permutation(K).astype(np.int16)
    tq, nq = _median_time(lambda: O.reorder_quant(xq, idx, "sim", 0.9), budget_s / 3, max_runs=3)
    return {"value": round(ops / t16 / 1e12, 4), "unit": "TOPS", "cores": os.cpu_count(), "threads": torch.get_num_threads(),
            "kind": "port",
            "sample": f"torch F.linear fp16 {M}x{N}x{K} on CPU (the literal call of the reference's QLinearLayer.forward), "
                      f"median of {n16} runs, {t16 * 1e3:.1f} ms each",
            "fp32_upcast": {"value": round(ops / t32 / 1e12, 4), "unit": "TOPS",
                            "sample": f"same operands upcast to fp32 (\"best CPU\"), median of {n32} runs, {t32 * 1e3:.1f} ms each"},
            "act_quant": {"value": round(Mq / tq, 1), "unit": "tokens/s",
                          "sample": f"numpy restatement of quantize_activation_wrapper (reorder + INT4/INT8 fake-quant), "
                                    f"{Mq} tokens x hidden {K}, 1 thread, median of {nq} runs, {tq * 1e3:.1f} ms each"}}


def max_over_ranks(values, dist, device):
    """Element-wise MAX of a list of per-rank floats over all ranks (identity when not distributed)."""
    if dist is None:
        return list(values)
    tt = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return tt.tolist()


def aggregate(world, steps, wall_s, kern_ms, M, N, K):
    """Whole-job numbers from the slowest rank's times: every rank ran `steps` independent replicas of the GEMM."""
    ops_per_step = 2.0 * M * N * K
    return {"value": world * ops_per_step * steps / wall_s / 1e12,
            "gbps": world * algorithmic_bytes(M, N, K) * steps / wall_s / 1e9,
            "ms_per_step": wall_s / steps * 1e3,
            "achieved": ops_per_step / (kern_ms * 1e-3) / 1e12}


def measured_traffic(M, N, K, fmt):
    """HBM bytes per launch of the headline kernel from the committed rocprofv3 PMC passes
    (profiles/rNN/bench_hbm_traffic*.json) -- only from a pass taken on the kernel sources of THIS tree
    (kernel_source_sha256 in the file, written by tools/summarize_profile.py), else None."""
    best = None
    sha = kernel_source_sha()
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for d in sorted(os.listdir(pdir)):
            for name in ("bench_hbm_traffic.json", f"bench_hbm_traffic_{fmt}.json"):
                f = os.path.join(pdir, d, name)
                if os.path.isfile(f):
                    t = json.load(open(f))
                    if t.get("algorithmic_bytes_per_launch") == algorithmic_bytes(M, N, K) and \
                            t.get("operand_format", "packed") == fmt and t.get("kernel_source_sha256") == sha:
                        best = int(t["traffic_bytes_per_launch"])
    return best


def build_f6_operands(ops_, M, N, K, dev):
    """The headline operands in the native F6 format (include/atom_hip.h, ATOM_AB_F6): activations as the fused quantisers
    emit them with ATOM_QUANT_F6_CODES (built here with torch from the same packed codes and scales), weights through
    atom_repack_weight_f6 (offline, once per weight)."""
    from atom_amd import ops as aops
    K4 = K - 128
    G = K4 // 128
    a4 = ops_[0].view(torch.uint8)
    nib = torch.stack([a4 & 0xF, a4 >> 4], dim=-1).reshape(M, K4).to(torch.int64)          # element 2j in the low nibble
    lut = torch.tensor([0x00, 0x0C, 0x10, 0x12, 0x14, 0x15, 0x16, 0x17, 0x38, 0x37, 0x36, 0x35, 0x34, 0x32, 0x30, 0x2C],
                       dtype=torch.int64, device=dev)                                      # BF6 (E3M2) code of nibble n
    c6 = lut[nib].reshape(M, G, 32, 4)
    w24 = c6[..., 0] | (c6[..., 1] << 6) | (c6[..., 2] << 12) | (c6[..., 3] << 18)
    by = torch.stack([(w24 >> (8 * i)) & 0xFF for i in range(3)], dim=-1).to(torch.uint8).reshape(M, G, 96)
    a6 = torch.zeros((G, aops.f6_rows(M), 104), dtype=torch.uint8, device=dev)
    a6[:, :M, :96] = by.permute(1, 0, 2)
    a6[:, :M, 96:98] = ops_[2].reshape(G, M, 1).view(torch.uint8).reshape(G, M, 2)         # sA[g, m] rides in the row: fp16 ...
    a6[:, :M, 100:104] = ops_[2].float().reshape(G, M, 1).view(torch.uint8).reshape(G, M, 4)   # ... and the same value as fp32
    b6 = aops.repack_weight_f6(ops_[1].view(torch.uint8), ops_[3])                         # codes + float32 scales (ATOM_B_F6S)
    return a6, b6


def cold_launches(step, dev, n=30, flush_mb=512):
    """The reference's second methodology (kernels/baselines/python-api.ipynb:33-56: one event pair per launch, an L2-sized memset
    between launches), sized for this chip: 512 MB zero-filled between launches evicts the 32 MB of L2 AND the 256 MB Infinity
    Cache, so every operand byte of the timed launch comes from HBM.  Returns (mean us, median us, n)."""
    flush = torch.empty(flush_mb << 20, dtype=torch.uint8, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = []
    for i in range(n + 3):
        flush.zero_()
        torch.cuda.synchronize(dev)
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize(dev)
        if i >= 3:
            times.append(e0.elapsed_time(e1) * 1e3)
    del flush
    return float(np.mean(times)), float(np.median(times)), len(times)


def configs_sweep(dev):
    """The other BASELINE configs beside the headline (N = 1 only; each row a fraction of a second): configs[1] (M = 1, N = K = 4096:
    the bandwidth-bound decode GEMM), the corners of configs[4] (hidden 5120: M in {1, 64, 256, 2048} x the two wide shapes) and the
    4096-wide batch sweep of the reference's own NVBench driver (bench_dense_layer_gemm_i4_o16.cu:64-69).  Packed (reference-format)
    operands through atom_gemm_w4a4_f16_ws, timed per launch by HIP-graph replay (the Python launch loop is slower than a 3 us kernel):
    hot = one operand set replayed, cold = the graph cycles through enough distinct weight sets (>= 600 MB) that every launch streams
    its weights from HBM -- what a model's layers do (tools/cold_bench.py).  From 129 rows the native BF6 operand format is timed
    beside it (`f6_us`, hot).  Fractions: of 8 TB/s on the algorithmic bytes (`frac_hbm`), of the dense INT8 MFMA peak (`frac_mfma`)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import cold_bench as CB
    from atom_amd import _lib as L
    lib = L.lib()
    rows = []
    shapes = [("C2", 1, 4096, 4096)]
    shapes += [("C5", m, n, k) for (n, k) in ((13824, 5120), (5120, 13824)) for m in (1, 64, 256, 2048)]
    shapes += [("C5", 64, 5120, 5120)]
    shapes += [("M-sweep", m, 4096, 4096) for m in (16, 64, 128, 256, 512, 1024)]
    for tag, M, N, K in shapes:
        hot, cold = CB.gemm_row(M, N, K, quiet=True)
        by, op = algorithmic_bytes(M, N, K), 2.0 * M * N * K
        row = {"config": tag, "M": M, "N": N, "K": K, "hot_us": round(hot, 2), "cold_us": round(cold, 2),
               "cold_gbps": round(by / cold / 1e3, 1), "cold_frac_hbm": round(by / cold / 1e6 / 8.0, 4),
               "hot_frac_hbm": round(by / hot / 1e6 / 8.0, 4),
               "cold_tops": round(op / cold / 1e6, 1), "cold_frac_mfma": round(op / cold / 1e6 / PEAK_I8_TOPS, 4)}
        if lib.atom_gemm_w4a4_ws_recodes_cached(M, N, K):      # the C ABI with one workspace per weight (ATOM_WS_WEIGHT_CACHED): activation re-coded per call
            wh, wc = CB.gemm_row(M, N, K, quiet=True, weight_cached=True)
            row.update({"wcached_hot_us": round(wh, 2), "wcached_cold_us": round(wc, 2)})
        if M >= 129:                                            # (the drop-in modules ask the quantisers for BF6 codes from 129 rows)
            ops_ = make_operands(M, N, K, dev, seed=1)
            a6, b6 = build_f6_operands(ops_, M, N, K, dev)
            D = torch.empty((M, N), dtype=torch.float16, device=dev)
            ptrs = [a6.data_ptr(), b6.data_ptr()] + [t.data_ptr() for t in ops_[2:]]
            f6_flags = L.SCALE_LAYOUT_PLAIN | L.AB_F6 | L.B_F6S | (L.B_SCALE_PAIRS if b6.atom_pairs else 0)
            f = lambda st: lib.atom_gemm_w4a4_f16(*ptrs, D.data_ptr(), M, N, K, 128, 128, f6_flags, st)
            f6 = CB.graph_time([f], 64)
            row.update({"f6_us": round(f6, 2), "f6_tops": round(op / f6 / 1e6, 1), "f6_frac_mfma": round(op / f6 / 1e6 / PEAK_I8_TOPS, 4)})
            if M <= 1024:                                       # ... and with the BF6 weights streamed from HBM: distinct copies >= 600 MB cycled
                wb = b6.atom_f6s.numel()
                copies = [b6.atom_f6s] + [b6.atom_f6s.clone() for _ in range(max(1, -(-CB.COLD_BYTES // wb)) - 1)]
                fs = []
                for w in copies:
                    pw = [a6.data_ptr(), w.data_ptr()] + [t.data_ptr() for t in ops_[2:]]
                    fs.append(lambda st, pw=pw: lib.atom_gemm_w4a4_f16(*pw, D.data_ptr(), M, N, K, 128, 128, f6_flags, st))
                n_it = max(64, 2 * len(fs))
                row["f6_cold_us"] = round(CB.graph_time(fs, n_it - n_it % len(fs)), 2)
                del copies, fs
            del ops_, a6, b6, D
        rows.append(row)
        torch.cuda.empty_cache()
    return {"method": "HIP-graph replay, per launch; hot = one operand set, cold = distinct weight sets >= 600 MB cycled (every launch "
                      "streams its weights from HBM); packed reference-format operands through atom_gemm_w4a4_f16_ws, nothing cached; "
                      "wcached_* = the same with one workspace per weight and ATOM_WS_WEIGHT_CACHED (its weight region filled once by "
                      "atom_repack_weight_f6s: only the activation is re-coded per call); f6_* = native BF6 operands",
            "rows": rows}


def kernel_source_sha():
    """sha256 over the kernel sources: a committed PMC pass is quoted only for the sources it was taken from."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "atom_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def spawn_ranks(n, argv):
    """--gpus N without a torch.distributed environment: start the N ranks the way the driver would."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def timed_steps(step, steps, warmup, sync, dist, device):
    """The contract's timed region: W untimed steps, then EXACTLY K steps bracketed by barrier + device synchronise on both
    sides; HIP events on the launch stream for the kernel-only time.  Returns (wall seconds, ms per step by events) as the MAX
    over ranks."""
    for _ in range(warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    cuda = device.type == "cuda"
    if cuda:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync()
    t0 = time.perf_counter()
    if cuda:
        e0.record()
    for _ in range(steps):
        step()
    t1 = time.perf_counter()
    if cuda:
        e1.record()
    sync()
    t2 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    kern_ms = (e0.elapsed_time(e1) if cuda else wall * 1e3) / steps
    timed_steps.breakdown = {"enqueue_us": round((t1 - t0) * 1e6, 1), "drain_us": round((t2 - t1) * 1e6, 1)}
    return max_over_ranks([wall, kern_ms], dist, device)


def block_workload(args, rank, world, dev):
    """BASELINE configs[3]: one Llama-7B decoder block W4A4 (g128, keeper 128, KV INT4), batch 32 x seq 2048 = 65,536 tokens,
    through atom_amd.model.qLlamaLayer.QLlamaDecoderLayer (the drop-in mirror of model/qLlamaLayer.py:86-127); every rank runs an
    independent replica.  A step = one block forward; `value` = TOPS of the seven W4A4 GEMMs inside it (timed as the four attention
    projections + the MLP module, whose gate / up / SiLU x up / quantiser are one launch; their share of the block time is in the line), roofline = those GEMMs against the INT8 MFMA peak.  cpu_baseline: the same seven projections as the
    reference computes them (F.linear on fp16 fake-quant operands) on the host at batch 1."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import block_bench
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)
    steps = min(args.steps, 10)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    r = block_bench.run(bsz=32, seq=2048, iters=steps, warmup=max(1, min(args.warmup, 3)), verbose=False, check=True)
    torch.cuda.synchronize(dev)
    # beside it: the same block with the attention opt-in of the configuration namespace (args.attn_sdpa: torch's fused attention
    # instead of the reference's materialised 32 x 32 x 2048 x 2048 score tensor -- same mathematics, not part of the W4A4 hot path)
    torch.cuda.empty_cache()
    r2 = block_bench.run(bsz=32, seq=2048, iters=steps, warmup=1, verbose=False, attn_sdpa=True)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    block_ms, gemm_ms = max_over_ranks([r["block_ms"], r["gemm_ms"]], dist, dev)
    if rank == 0:
        tops = world * r["gemm_ops"] / (gemm_ms * 1e-3) / 1e12
        out = {"metric": "effective TOPS of the seven W4A4 GEMMs inside one Llama-7B block (batch 32 x seq 2048, KV INT4)",
               "value": round(tops, 2), "unit": "TOPS", "n_gpus": world, "steps": steps, "warmup": max(1, min(args.warmup, 3)),
               "ms_per_step": round(block_ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "int4xint4 exact integer dot, fp32 dequant, fp16 out; attention / RoPE / residuals fp16 (torch)",
               "data": "synthetic",
               "config": {"workload": "Llama-7B decoder block W4A4 g128 keeper 128, batch 32 x seq 2048, KV-cache INT4 fake-quant "
                                      "(BASELINE configs[3]) through QLlamaDecoderLayer", "tokens": r["tokens"],
                          "parallelism": f"replicas x{world}"},
               "block_ms": round(block_ms, 3), "gemm_ms": round(gemm_ms, 3), "gemm_share": round(gemm_ms / block_ms, 4),
               "module_ms": r["spans"],
               # the timed model is the model of tests/golden/llama_block_7b_wide.npz: its seven projections on the unmodified reference's
               # own inputs (256 token rows) against the reference's own outputs (256 features each) -- tools/block_bench.py golden_check
               "golden_check": {"rows": 256, "features_per_projection": 256,
                                "rel_frobenius": {k: float(f"{v[0]:.3e}") for k, v in r.get("golden", {}).items()},
                                "worst_element": {k: float(f"{v[1]:.3e}") for k, v in r.get("golden", {}).items()},
                                "bounds": {"rel_frobenius": 1e-3, "worst_element": 1e-2},
                                "ok": bool(r.get("golden")) and all(v[0] <= 1e-3 and v[1] <= 1e-2 for v in r["golden"].values())},
               "block_ms_attn_sdpa": round(r2["block_ms"], 3), "gemm_ms_attn_sdpa": round(r2["gemm_ms"], 3),
               "roofline": {"bound": "mfma", "achieved": round(r["gemm_ops"] / (gemm_ms * 1e-3) / 1e12, 2), "peak": PEAK_I8_TOPS,
                            "unit": "TFLOP/s", "frac": round(r["gemm_ops"] / (gemm_ms * 1e-3) / 1e12 / PEAK_I8_TOPS, 4),
                            "traffic": None, "algorithmic_ops": int(r["gemm_ops"])},
               "timed_region_s": round(time.perf_counter() - t0, 2)}
        if world == 1 and not args.no_cpu_baseline:
            g = torch.Generator().manual_seed(0)
            shapes = [(4096, 4096)] * 4 + [(11008, 4096)] * 2 + [(4096, 11008)]
            xs = {k: (torch.randn(2048, k, generator=g) * 0.5).half() for k in (4096, 11008)}
            ws = [(torch.randn(n, k, generator=g) * 0.05).half() for n, k in shapes]
            from oracle import atom_oracle as O
            t, n = _median_time(lambda: [O.sim_linear_torch(xs[w.shape[1]], w) for w in ws], 12.0, max_runs=3)
            ops1 = 2.0 * 2048 * sum(n_ * k_ for n_, k_ in shapes)
            out["cpu_baseline"] = {"value": round(ops1 / t / 1e12, 4), "unit": "TOPS", "cores": os.cpu_count(),
                                   "threads": torch.get_num_threads(), "kind": "port",
                                   "sample": f"the seven projections of the block at batch 1 x seq 2048 as the reference computes them "
                                             f"(F.linear on fp16 operands, torch CPU kernels), median of {n} runs, {t:.2f} s each"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--ramp", type=int, default=1500, help="untimed launches before the warm-up steps (power-state ramp); "
                                                           "reported in the JSON line")
    ap.add_argument("--format", choices=["f6", "packed", "wide"], default="f6",
                    help="operand format of the headline measurement (the other two are reported beside it at N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the side sweep over the other BASELINE configs (N=1 only)")
    ap.add_argument("--with-block", action="store_true", help="(the default since round 5; kept so that older command lines parse)")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode-layer key (tools/cold_bench.py layer 1,16 in its own process, ~20 s)")
    ap.add_argument("--no-block", action="store_true", help="skip --workload block (BASELINE configs[3]: one Llama-7B block at batch 32 x "
                                                              "seq 2048, its own process; ~1 minute), whose numbers otherwise go into "
                                                              "the line's `block` key (N=1 only)")
    ap.add_argument("--workload", choices=["gemm", "block"], default="gemm",
                    help="gemm = the headline GEMM (BASELINE configs[2]); block = one Llama-7B decoder block, batch 32 x seq 2048, "
                         "KV INT4, through QLlamaDecoderLayer (BASELINE configs[3]; a step = one block forward; --steps 3 is plenty)")
    ap.add_argument("--stub-step", action="store_true",
                    help="tests only: the launch / rendezvous / timing / aggregation path on CPU (gloo) with a dummy step")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)")
    M, N, K = args.M, args.N, args.K

    if args.stub_step:                                           # CPU stand-in for the kernel launch (tests/test_bench_gloo.py)
        dev = torch.device("cpu")
        dist = None
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group(backend="gloo")
        acc = torch.zeros(64, 64)
        eye = torch.eye(64)

        def step():
            acc.add_(eye @ eye)
        wall, kern_ms = timed_steps(step, args.steps, args.warmup, lambda: None, dist, dev)
        if rank == 0:
            agg = aggregate(world, args.steps, wall, kern_ms, M, N, K)
            print(json.dumps({"metric": "stub", "value": agg["value"], "unit": "TOPS", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": agg["ms_per_step"], "ramp": 0, "data": "stub step on CPU",
                              "steps_run": int(acc[0, 0].item())}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the W4A4 path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.workload == "block":
        return block_workload(args, rank, world, dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)

    from atom_amd import _lib as L
    lib = L.lib()
    ops_ = make_operands(M, N, K, dev, seed=rank)
    D = torch.empty((M, N), dtype=torch.float16, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ptrs = [t.data_ptr() for t in ops_]

    # the three operand formats of the same GEMM (same codes, same scales, bit-identical results)
    a4 = ops_[0].view(torch.uint8).view(M, (K - 128) // 32, 16)
    wide = torch.stack([(a4 << 4) & 0xF0, a4 & 0xF0], dim=2).reshape(M, K - 128).contiguous()
    a6, b6 = build_f6_operands(ops_, M, N, K, dev)
    # channel pairs share their weight scales in these operands (make_operands; weight_channel_group = 2, the reference kernel's only
    # form): the callers say so, as a binding of the reference's launcher would (ATOM_B_SCALE_PAIRS; atom_amd.ops checks the values)
    PAIRS = L.B_SCALE_PAIRS if b6.atom_pairs else 0
    variants = {
        "packed": ([*ptrs], L.SCALE_LAYOUT_PLAIN | PAIRS,
                   "reference format: packed INT4 nibbles (punica.ops ABI), INT8 MFMA after in-kernel widening"),
        "wide": ([wide.data_ptr()] + ptrs[1:], L.SCALE_LAYOUT_PLAIN | L.A_WIDE | PAIRS,
                 "activations int8 = code*16 (ATOM_A_WIDE), weights packed INT4, INT8 MFMA"),
        "f6": ([a6.data_ptr(), b6.data_ptr()] + ptrs[2:], L.SCALE_LAYOUT_PLAIN | L.AB_F6 | L.B_F6S | PAIRS,
               "both operands BF6 group-major (ATOM_AB_F6): v_mfma_f32_16x16x128_f8f6f4, exact integer dot products; the "
               "format the fused quantisers emit for prefill batches"),
        "packed_ws": ([*ptrs], L.SCALE_LAYOUT_PLAIN | PAIRS,
                      "reference packed format through atom_gemm_w4a4_f16_ws with a caller-owned workspace, nothing cached: BOTH operands "
                      "are re-coded to BF6 in the workspace by every call, then the BF6 MFMA kernel -- what a one-to-one binding of the "
                      "reference's launcher with a workspace gets (INTEGRATION.md)"),
        "packed_ws_same_weight": ([*ptrs], L.SCALE_LAYOUT_PLAIN | PAIRS,
                                  "as packed_ws, REPEATED CALLS WITH ONE WEIGHT: the caller asserts ATOM_WS_WEIGHT_CACHED from the second call "
                                  "on (only the activation is re-coded); holds only while nothing else touches that workspace"),
        "packed_ops": (None, None,
                       "reference packed format through atom_amd.ops.dense_layer_gemm_i4_fp16 (the punica.ops surface), TWO weights in "
                       "alternation so that a one-slot cache cannot help: ops keeps every weight's BF6 form in a per-weight LRU cache and "
                       "re-codes only the activation -- what a punica.ops caller walking a model's projections gets"),
    }
    ws_bytes = lib.atom_gemm_w4a4_workspace_bytes(M, N, K)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    ws_state = {"flag": 0}
    last = {"d": None}                                             # output of the ops-level variant (ops allocates its result)

    def make_step(name):
        vp, layout, _ = variants[name]
        if name == "packed_ops":
            from atom_amd import ops as aops
            second = [t.clone() for t in (ops_[1], ops_[3], ops_[5], ops_[7])]            # another weight (same values, its own storage)
            weights = [(ops_[1], ops_[3], ops_[5], ops_[7]), tuple(second)]
            turn = {"i": 0}

            def step():
                b, sb, b8, sb8 = weights[turn["i"] & 1]
                turn["i"] += 1
                last["d"] = aops.dense_layer_gemm_i4_fp16(ops_[0], b, ops_[2], sb, ops_[4], b8, ops_[6], sb8, scale_layout="plain")
            return step

        def step():
            if name.startswith("packed_ws"):
                flag = ws_state["flag"] if name == "packed_ws_same_weight" else 0
                st = lib.atom_gemm_w4a4_f16_ws(*vp, D.data_ptr(), M, N, K, 128, 128, layout | flag, ws.data_ptr(), ws_bytes, stream)
                ws_state["flag"] = L.WS_WEIGHT_CACHED if lib.atom_gemm_w4a4_ws_recodes(M, N, K) else 0
            else:
                st = lib.atom_gemm_w4a4_f16(*vp, D.data_ptr(), M, N, K, 128, 128, layout, stream)
            if st != 0:
                L.check(st, "atom_gemm_w4a4_f16")
        return step

    step = make_step(args.format)
    sync = lambda: torch.cuda.synchronize(dev)
    # Power-state ramp, part of the set-up (not of the W warm-up steps, not timed, reported as "ramp"): after idle the chip
    # runs the first few hundred launches at lower clocks (500 timed steps after 100 warm-ups measure ~5 % below 2000 after
    # 200), so bring it to its sustained state before the contract's W untimed + K timed steps.
    for _ in range(args.ramp):
        step()
    if args.ramp:                                # ... and one dry pass through the timing apparatus itself (event creation, first
        timed_steps(step, 2, 0, sync, dist, dev)  # synchronize / barrier of the process): 2 more untimed launches, counted in "ramp"
    wall, kern_ms = timed_steps(step, args.steps, args.warmup, sync, dist, dev)

    # N=1 only: the headline kernel with every operand byte coming from HBM ("additionally reported with a 512 MB flush
    # buffer"), then the other operand formats beside the headline, and a bit-for-bit comparison of the outputs
    others = {}
    cold = None
    if world == 1:
        c_mean, c_med, c_n = cold_launches(step, dev)
        cold = {"kernel_us": round(c_mean, 2), "median_us": round(c_med, 2), "launches": c_n,
                "value": round(2.0 * M * N * K / (c_mean * 1e-6) / 1e12, 2), "unit": "TOPS",
                "frac": round(2.0 * M * N * K / (c_mean * 1e-6) / 1e12 / PEAK_I8_TOPS, 4),
                "method": "one event pair per launch, 512 MB zero-filled between launches (L2 32 MB + Infinity Cache 256 MB evicted; "
                          "kernels/baselines/python-api.ipynb:33-56 scaled to this chip); the launch starts from an idle, flushed chip"}
        for _ in range(200):                                     # back to the sustained state for the side measurements
            step()
        D_head = D.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for name in variants:
            if name == args.format:
                continue
            st2 = make_step(name)
            D.zero_()
            # side measurements, not the contract's K steps: each format gets its own ramp like the headline's (its kernels draw
            # different power and the chip takes several hundred launches to settle: 20 steps after 5 launches read 75 us for
            # packed_ws, 100 after 100 read 67, sustained 65; the INT8 kernel 106 / 109 / 94) and at least 100 timed steps; both
            # counts are reported
            n_warm, n_steps = max(args.warmup, min(args.ramp, 1000) if args.ramp else 100), max(args.steps, 100)
            for _ in range(n_warm):                              # (at least one launch: the comparison below is of ITS output)
                st2()
            torch.cuda.synchronize(dev)
            same = bool(torch.equal(last["d"] if name == "packed_ops" else D, D_head))
            e0.record()
            for _ in range(n_steps):
                st2()
            e1.record()
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / n_steps
            tops = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            others[name] = {"value": round(tops, 2), "unit": "TOPS", "kernel_us": round(ms * 1e3, 2),
                            "frac": round(tops / PEAK_I8_TOPS, 4), "steps": n_steps, "warmup": n_warm,
                            "bit_identical_to_headline_output": same,
                            "format": variants[name][2]}

    if rank == 0:
        agg = aggregate(world, args.steps, wall, kern_ms, M, N, K)
        ops_per_step = 2.0 * M * N * K
        tops, ach = agg["value"], agg["achieved"]
        out = {
            "metric": "effective TOPS, W4A4 group-128 GEMM + 128 INT8 outlier cols, fp16 out",
            "value": round(tops, 2), "unit": "TOPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(agg["ms_per_step"], 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int4xint4 exact integer dot (BF6-coded on the f8f6f4 MFMA; INT8 MFMA for the keeper), fp32 dequant, fp16 out"
                     if args.format == "f6" else "int4xint4->int32 (i8 MFMA), fp32 dequant, fp16 out",
            "data": "synthetic",
            "timed_region": {"wall_us": round(wall * 1e6, 1), "events_us": round(kern_ms * 1e3 * args.steps, 1),
                             **getattr(timed_steps, "breakdown", {})},
            "ramp": args.ramp + (2 if args.ramp else 0),
            "ramp_note": "untimed set-up launches of the same kernel ahead of the W warm-up steps (power-state ramp after idle)",
            "config": {"workload": f"W4A4 GEMM M={M} N={N} K={K} group=128 keeper=128 (BASELINE configs[2])",
                       "M": M, "N": N, "K": K, "parallelism": f"replicas x{world}", "operand_format": args.format,
                       "operand_format_note": variants[args.format][2]},
            "gbps": round(agg["gbps"], 1),
            "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_I8_TOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_I8_TOPS, 4), "traffic": measured_traffic(M, N, K, args.format),
                         "traffic_note": "HBM bytes per launch from the committed rocprofv3 PMC passes of this command on these kernel "
                                         "sources (profiles/rNN/bench_hbm_traffic_*.json; null when no pass of the same sources is "
                                         "committed), not measured in this run",
                         "kernel_us": round(kern_ms * 1e3, 2),
                         "algorithmic_bytes": algorithmic_bytes(M, N, K), "algorithmic_ops": int(ops_per_step),
                         "peak_note": "dense INT8 MFMA peak (BASELINE.md); the BF6 MFMA this format runs on peaks at ~10 POPS"},
        }
        if cold is not None:
            out["cold"] = cold
        if others:
            out["other_operand_formats"] = others
            if "packed_ops" in others:                           # what a punica.ops caller (the reference's operand ABI) gets
                out["abi_value"] = others["packed_ops"]["value"]
                out["abi_value_note"] = ("TOPS of the same GEMM called with the reference's packed operand format through atom_amd.ops, two "
                                         "weights in alternation (packed_ops); packed_ws = nothing cached, "
                                         "packed_ws_same_weight = the round-3 figure (one weight repeated)")
        if world == 1 and not args.no_configs:
            del a6, b6, wide
            torch.cuda.empty_cache()
            out["configs"] = configs_sweep(dev)
        if world == 1 and not args.no_block:                     # BASELINE configs[3] at its stated size in the same line (its own
            import subprocess                                    # process: fresh allocator state; ~1 minute)
            torch.cuda.empty_cache()
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "block", "--steps", "3", "--warmup", "1",
                                "--no-cpu-baseline"], capture_output=True, text=True)
            try:
                b = json.loads(r.stdout.strip().splitlines()[-1])
                mm = b["module_ms"]
                quant_ms = sum(v for k, v in mm.items() if k.endswith("layernorm"))
                out["block"] = {"config": b["config"]["workload"], "tokens": b["config"]["tokens"], "steps": b["steps"],
                                "block_ms": b["block_ms"],
                                "gemm_ms": b["gemm_ms"], "gemm_tops": b["value"], "gemm_frac_mfma": b["roofline"]["frac"],
                                "quantiser_ms": round(quant_ms, 3),
                                "attention_and_rest_ms": round(b["block_ms"] - b["gemm_ms"] - quant_ms, 3),
                                "gemm_share": b["gemm_share"], "module_ms": mm,
                                "block_ms_attn_sdpa": b.get("block_ms_attn_sdpa"), "gemm_ms_attn_sdpa": b.get("gemm_ms_attn_sdpa"),
                                "golden_check": b.get("golden_check"),
                                "note": "one QLlamaDecoderLayer forward (model/qLlamaLayer.py:86-127 mirrored) at batch 32 x seq 2048; gemm_ms = "
                                        "q/k/v/o projections + the MLP module (gate / up / SiLU x up / its quantiser in one launch, then down_proj); "
                                        "quantiser_ms = the two RMSNorm -> reorder -> quantise launches (the o_proj and down_proj quantisers run "
                                        "inside attention / the MLP launch); the rest is torch's attention over the fake-quantised INT4 KV -- the "
                                        "reference's materialised score matrix, model/qLlamaLayer.py:262-290 --, RoPE and the residual adds; "
                                        "*_attn_sdpa = the same block with the opt-in args.attn_sdpa (torch's fused attention)"}
            except Exception as e:                               # pragma: no cover
                out["block"] = {"error": f"{type(e).__name__}: {e}; stderr tail: {r.stderr[-300:]}"}
        if world == 1 and not args.no_decode:                    # the decode step of a Llama-7B layer (reference punica/models/llama.py:259-292):
            import subprocess                                    # its own process too (tools/cold_bench.py: HIP-graph replay, hot / cold)
            torch.cuda.empty_cache()
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cold_bench.py"), "layer", "1,16"], capture_output=True, text=True)
            try:
                rows = {}
                for ln in r.stdout.splitlines():
                    if ln.startswith("batch"):
                        f = ln.split()
                        rows[f"batch_{int(f[1].rstrip(':'))}"] = {"hot_us": float(f[3]), "cold_us": float(f[6]),
                                                                  "cold_frac_hbm": round(12.6 / float(f[6]), 3)}
                assert rows
                out["decode_layer"] = {"config": "Llama-7B decoder layer (atom_amd.e2e.LlamaDecoderLayer), one decode step, context 1024, INT4 "
                                                 "paged KV, HIP-graph replay, us per layer", **rows,
                                       "weight_stream_us_at_8TBps": 12.6,
                                       "note": "hot = one layer replayed (its 101 MB of weights stay in the Infinity Cache); cold = 8 distinct "
                                               "layers per replay, every launch streams its weights from HBM; cold_frac_hbm = 12.6 us (the "
                                               "layer's weight bytes at 8 TB/s) / cold_us.  Batch 1: five launches since round 6 (the four "
                                               "activation quantisers inside their projections' launches, csrc/gemvq_w4a4.hip; KV quant + "
                                               "append inside the attention launch; the KV-split merge inside o_proj's) -- round 5: ten, "
                                               "55.3 / 59.7 us"}
            except Exception as e:                               # pragma: no cover
                out["decode_layer"] = {"error": f"{type(e).__name__}: {e}; stderr tail: {r.stderr[-300:]}"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(M, N, K)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
