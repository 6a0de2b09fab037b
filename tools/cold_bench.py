"""Hot vs COLD timing of the decode-size GEMMs: a serving step streams every layer's weights once, so a
number measured by replaying ONE operand set (8.9 MB at M = 1, N = K = 4096) out of the 32 MB of L2 / 256 MB of Infinity Cache is a
cache figure, not an HBM figure.

    hot   the same operand set in every launch of the captured HIP graph (what tools/r02/decode_probe.py and sweep.py measured)
    cold  the graph cycles through R distinct weight sets with R x (weight bytes) >= 600 MB (> L2 + Infinity Cache), so every launch
          reads weights that were evicted since their last use; activations / outputs (KBs) are shared.  Per-launch time = graph
          replay time / launches, exactly as in the hot case -- same launch floor, same graph.

    python tools/cold_bench.py [gemm|layer|all]

gemm:  M in {1, 2, 8, 16, 32, 64, 128, 256} x the Llama-7B / 13B projection shapes, through atom_gemm_w4a4_f16_ws (what atom_amd.ops calls).
layer: one decode step of a Llama-7B layer (atom_amd.e2e.LlamaDecoderLayer), hot = one layer replayed, cold = 8 distinct layers
       (8 x 101 MB of INT4 weights) in one graph.
"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from atom_amd import _lib as L  # noqa: E402

if os.environ.get("ATOM_LIB"):                 # tools only: time another build of the library (tools/ab_build.sh) on the same box
    L.LIB_PATH = os.path.abspath(os.environ["ATOM_LIB"])
dev = torch.device("cuda", 0)
lib = L.lib()
COLD_BYTES = 600 << 20


def graph_time(launchers, iters):
    """launchers: list of callables taking the capture stream; `iters` launches are captured cycling through them."""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        st = torch.cuda.current_stream().cuda_stream
        for f in launchers[:3]:
            f(st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        st = torch.cuda.current_stream().cuda_stream
        for i in range(iters):
            launchers[i % len(launchers)](st)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


WIDE = bool(int(os.environ.get("ATOM_BENCH_WIDE", "0")))      # tools only: pre-widened activations (ATOM_A_WIDE) instead of packed nibbles


def widen(a4):
    """packed nibbles [M, K4/2] -> the wide activation format int8 [M, K4]: per 32 channels 16 even then 16 odd codes, x 16."""
    b = a4.view(torch.uint8).to(torch.int16)
    even = ((b & 0xF) << 4).to(torch.uint8).view(torch.int8)       # channel 2i  -> byte i (two's complement nibble x 16)
    odd = (b & 0xF0).to(torch.uint8).view(torch.int8)              # channel 2i+1
    M = a4.shape[0]
    return torch.stack([even.reshape(M, -1, 16), odd.reshape(M, -1, 16)], 2).reshape(M, -1).contiguous()


def gemm_sets(M, N, K, R):
    """R operand sets sharing the activation side (a4, sA, a8, sA8) and the output; distinct weights."""
    base = list(bench.make_operands(M, N, K, dev, seed=1))
    if WIDE:
        base[0] = widen(base[0])
    sets = [base]
    for r in range(1, R):
        w = [torch.empty_like(base[i]).copy_(base[i]) for i in (1, 3, 5, 7)]     # b4, sB, b8, sB8: distinct memory, same values
        sets.append([base[0], w[0], base[2], w[1], base[4], w[2], base[6], w[3]])
    return sets


def gemm_row(M, N, K, quiet=False, weight_cached=False):
    """weight_cached: the caller keeps ONE workspace per weight, its weight region filled once by atom_repack_weight_f6s, and passes
    ATOM_WS_WEIGHT_CACHED (include/atom_hip.h) -- what a binding that owns its layers does; only the activation is re-coded per call."""
    wbytes = N * (K - 128) // 2 + N * 128 + 2 * N * ((K - 128) // 128 + 1)
    R = max(2, -(-COLD_BYTES // wbytes))
    sets = gemm_sets(M, N, K, R)
    D = torch.empty((M, N), dtype=torch.float16, device=dev)
    wsb = lib.atom_gemm_w4a4_workspace_bytes(M, N, K)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    flags = L.SCALE_LAYOUT_PLAIN | L.B_SCALE_PAIRS | (L.A_WIDE if WIDE else 0)   # (bench.make_operands: channel pairs share their scales)
    cached = weight_cached and wsb > 0 and lib.atom_gemm_w4a4_packed_order(M, N, K, 2) in (1, 2, 4)

    def mk(ops_):
        ptrs = [t.data_ptr() for t in ops_]
        if cached:
            w = torch.empty(wsb, dtype=torch.uint8, device=dev)
            L.check(lib.atom_repack_weight_f6s(ptrs[1], ptrs[3], N, K, w.data_ptr(), None), "atom_repack_weight_f6s")
            return lambda st, w=w: lib.atom_gemm_w4a4_f16_ws(*ptrs, D.data_ptr(), M, N, K, 128, 128, flags | L.WS_WEIGHT_CACHED, w.data_ptr(), wsb, st)
        return lambda st: lib.atom_gemm_w4a4_f16_ws(*ptrs, D.data_ptr(), M, N, K, 128, 128, flags, ws.data_ptr(), wsb, st)
    launchers = [mk(s) for s in sets]
    torch.cuda.synchronize()
    iters = max(64, 2 * R)
    iters -= iters % R
    hot = graph_time(launchers[:1], iters)
    cold = graph_time(launchers, iters)
    by = bench.algorithmic_bytes(M, N, K)
    if not quiet:
        print(f"{M:5d} {N:6d} {K:6d} | hot {hot:7.2f} us {by / hot / 1e6:5.2f} TB/s ({by / hot / 1e6 / 8:.3f}) | cold {cold:7.2f} us {by / cold / 1e6:5.2f} TB/s "
          f"({by / cold / 1e6 / 8:.3f} of 8 TB/s) | {R} weight sets x {wbytes / 1e6:.1f} MB", flush=True)
    return hot, cold


def gemm_main(ms=None):
    print("# decode-size W4A4 GEMMs, HIP-graph replay, per launch; (fraction of 8 TB/s on the algorithmic bytes)")
    print("#   M      N      K")
    for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008), (5120, 5120), (13824, 5120), (5120, 13824)):
        for M in (ms or ((1, 2, 8, 16, 32, 64, 128, 256) if (N, K) == (4096, 4096) else (1, 16, 64, 256))):
            gemm_row(M, N, K)


def layer_main(ctx=1024, batches=(1, 16, 64), hidden=4096, heads=32, inter=11008, nlayers=8):
    from atom_amd.e2e import LlamaDecoderLayer
    from atom_amd.e2e import llama as _e2e
    if "ATOM_FUSED_Q_MASK" in os.environ:                    # tools only: which quantisers ride inside their consumers (e2e.DecodeFusion)
        _e2e.FUSION.q_mask = int(os.environ["ATOM_FUSED_Q_MASK"])
    if "ATOM_MERGE_IN_O_PROJ" in os.environ:                 # ... the KV-split merge inside o_proj's launch
        _e2e.FUSION.merge_in_o_proj = bool(int(os.environ["ATOM_MERGE_IN_O_PROJ"]))
    if "ATOM_FUSED_Q_MASK2" in os.environ:                   # ... at two tokens
        _e2e.FUSION.q_mask2 = int(os.environ["ATOM_FUSED_Q_MASK2"])
    from atom_amd.utils import BatchLenInfo, BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4
    cfg = types.SimpleNamespace(hidden_size=hidden, num_attention_heads=heads, intermediate_size=inter, rms_norm_eps=1e-5, rope_theta=1e4)
    g = torch.Generator().manual_seed(1)
    if "ATOM_LAYER_DIMS" in os.environ:                       # tools only: another model's layer, "hidden,heads,inter" (Llama-13B: 5120,40,13824)
        hidden, heads, inter = (int(v) for v in os.environ["ATOM_LAYER_DIMS"].split(","))
        cfg = types.SimpleNamespace(hidden_size=hidden, num_attention_heads=heads, intermediate_size=inter, rms_norm_eps=1e-5, rope_theta=1e4)
    layers = []
    for li in range(nlayers):
        layer = LlamaDecoderLayer(cfg, layer_idx=0).cuda()
        for mod in layer.modules():
            if type(mod).__name__ == "LinearInt4":
                mod.load_fp16_weight((torch.randn(mod.out_features, mod.in_features, generator=g) * 0.05).half().cuda())
            elif type(mod).__name__ == "LlamaRMSNormInt4":
                mod.weight.data = (1 + 0.1 * torch.randn(mod.weight.shape, generator=g)).half().cuda()
        layers.append(layer)
    print(f"# Llama-7B decoder layer, one decode step, context {ctx}, INT4 paged KV; HIP-graph replay, us per layer")
    print(f"#   hot = one layer replayed {nlayers} times (its ~101 MB of weights stay in the Infinity Cache); cold = {nlayers} distinct layers "
          f"({nlayers * 101} MB of weights) per replay")
    for bsz in batches:
        pool = KvPoolInt4(num_layers=1, num_heads=heads, head_dim=hidden // heads, capacity=bsz * (ctx // 16 + 2), block_len=16, device=dev)
        pool.buf.random_(0, 255)
        pool.param.copy_(torch.rand(pool.param.shape, device=dev).mul_(0.05).add_(0.01).half())
        cs = [KvCacheInt4(pool, ctx) for _ in range(bsz)]
        for c in cs:
            c.acquire_one()
        kv = BatchedKvCacheInt4(cs)
        x = (torch.randn(bsz, hidden, device=dev) * 0.7).half()
        blen = BatchLenInfo([], bsz, dev)
        res = {}
        with torch.no_grad():
            for name, seq in (("hot", [layers[0]] * nlayers), ("cold", layers)):
                for l in seq:
                    l(x, blen, None, kv)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for l in seq:
                        l(x, blen, None, kv)
                gr.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                best = 1e9
                for _ in range(5):
                    e0.record()
                    for _ in range(4):
                        gr.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / 4 / nlayers)
                res[name] = best
        print(f"batch {bsz:4d}: hot {res['hot']:7.1f} us  cold {res['cold']:7.1f} us per layer   (weight stream at 8 TB/s: {101e6 / 8e12 * 1e6:.1f} us)", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("gemm", "all"):
        gemm_main()
    if what == "gemm_m":                                      # python tools/cold_bench.py gemm_m 2,16,32
        gemm_main(tuple(int(b) for b in sys.argv[2].split(",")))
    if what in ("layer", "all"):
        layer_main(batches=tuple(int(b) for b in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 16, 64))
