#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Atom W4A4 group-128 GEMM on MI355X.

One "step" = one launch of atom_gemm_w4a4_f16 (through the C ABI) on the headline workload of BASELINE.json
(configs[2]: M=N=K=4096, group 128, 128 INT8 outlier columns), operands resident in HBM, synthetic uniform random
int4/int8 codes and U(0.005,0.05) fp16 scales (never zeros: zero data clocks ~19 % higher).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--M 4096 --N 4096 --K 4096] [--no-cpu-baseline]

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


def cpu_baseline(M, N, K, budget_s=15.0):
    """The reference's CPU path for this GEMM: QLinearLayer.forward == F.linear on fake-quant FP16 operands
    (model/qLinearLayer.py:32-35), torch CPU kernels, all host threads.  Bounded sample."""
    from oracle import atom_oracle as O
    g = torch.Generator().manual_seed(0)
    Ms = M
    x = (torch.randn(Ms, K, generator=g) * 0.5).half()
    w = (torch.randn(N, K, generator=g) * 0.05).half()
    t0 = time.perf_counter()
    O.sim_linear_torch(x, w)                                     # warm-up
    first = time.perf_counter() - t0
    times = []
    start = time.perf_counter()
    while len(times) < 10 and (time.perf_counter() - start) < budget_s:
        t0 = time.perf_counter()
        O.sim_linear_torch(x, w)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times)) if times else first
    return {"value": round(2.0 * Ms * N * K / t / 1e12, 4), "unit": "TOPS", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"torch F.linear fp16 {Ms}x{N}x{K} on CPU (reference QLinearLayer.forward), "
                      f"median of {max(len(times), 1)} runs, {t * 1e3:.1f} ms each"}


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


def measured_traffic(M, N, K):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN/bench_hbm_traffic.json), or None."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for d in sorted(os.listdir(pdir)):
            f = os.path.join(pdir, d, "bench_hbm_traffic.json")
            if os.path.isfile(f):
                t = json.load(open(f))
                if t.get("algorithmic_bytes_per_launch") == algorithmic_bytes(M, N, K):
                    best = int(t["traffic_bytes_per_launch"])
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the W4A4 path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)

    from atom_amd import _lib as L
    lib = L.lib()
    M, N, K = args.M, args.N, args.K
    ops_ = make_operands(M, N, K, dev, seed=rank)
    D = torch.empty((M, N), dtype=torch.float16, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ptrs = [t.data_ptr() for t in ops_]

    def step():
        st = lib.atom_gemm_w4a4_f16(*ptrs, D.data_ptr(), M, N, K, 128, 128, L.SCALE_LAYOUT_PLAIN, stream)
        if st != 0:
            L.check(st, "atom_gemm_w4a4_f16")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1) / args.steps                  # HIP events on the launch stream
    wall, kern_ms = max_over_ranks([wall, kern_ms], dist, dev)

    # Second line of evidence (N=1 only, reported beside the headline, never as `value`): the same GEMM fed the native
    # activation format the fused quant kernels can emit (int8 code*16, ATOM_A_WIDE) instead of the reference's packed
    # nibbles -- the format the B1 modules use for prefill-sized batches.
    native = None
    if world == 1:
        a4 = ops_[0].view(torch.uint8).view(M, (K - 128) // 32, 16)
        wide = torch.stack([(a4 << 4) & 0xF0, a4 & 0xF0], dim=2).reshape(M, K - 128).contiguous()
        wptrs = [wide.data_ptr()] + ptrs[1:]

        def step_w():
            st = lib.atom_gemm_w4a4_f16(*wptrs, D.data_ptr(), M, N, K, 128, 128, L.SCALE_LAYOUT_PLAIN | L.A_WIDE, stream)
            if st != 0:
                L.check(st, "atom_gemm_w4a4_f16")

        for _ in range(args.warmup):
            step_w()
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(args.steps):
            step_w()
        e1.record()
        torch.cuda.synchronize(dev)
        wms = e0.elapsed_time(e1) / args.steps
        native = {"value": round(2.0 * M * N * K / (wms * 1e-3) / 1e12, 2), "unit": "TOPS", "kernel_us": round(wms * 1e3, 2),
                  "frac": round(2.0 * M * N * K / (wms * 1e-3) / 1e12 / PEAK_I8_TOPS, 4),
                  "format": "activations int8 = code*16 (ATOM_A_WIDE), weights packed INT4"}

    if rank == 0:
        agg = aggregate(world, args.steps, wall, kern_ms, M, N, K)
        ops_per_step = 2.0 * M * N * K
        tops, ach = agg["value"], agg["achieved"]
        out = {
            "metric": "effective TOPS, W4A4 group-128 GEMM + 128 INT8 outlier cols, fp16 out",
            "value": round(tops, 2), "unit": "TOPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(agg["ms_per_step"], 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int4xint4->int32 (i8 MFMA), fp32 dequant, fp16 out", "data": "synthetic",
            "config": {"workload": f"W4A4 GEMM M={M} N={N} K={K} group=128 keeper=128 (BASELINE configs[2])",
                       "M": M, "N": N, "K": K, "parallelism": f"replicas x{world}"},
            "gbps": round(agg["gbps"], 1),
            "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_I8_TOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_I8_TOPS, 4), "traffic": measured_traffic(M, N, K),
                         "kernel_us": round(kern_ms * 1e3, 2),
                         "algorithmic_bytes": algorithmic_bytes(M, N, K), "algorithmic_ops": int(ops_per_step)},
        }
        if native is not None:
            out["native_wide_activations"] = native
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(M, N, K)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
