"""Small-M latency of the activation quantisers under HIP-graph replay (decode steps run them at 1..16 rows): us per launch incl. the
launch floor (a trivial kernel costs ~1.65 us under the same replay)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from atom_amd import ops
dev = torch.device("cuda", 0)


def graph_time(fn, iters=100):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(5):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


x0 = torch.zeros(64, device=dev)
print("tiny torch kernel: %.2f us" % graph_time(lambda: x0.add_(1)))
for H in (4096, 11008):
    idx = torch.randperm(H, device=dev).to(torch.int16)
    w = (torch.randn(H, device=dev) * 0.1 + 1).half()
    for M in (1, 4, 16, 64, 256):
        x = torch.randn(M, H, device=dev).half()
        b = torch.randn(M, H, device=dev).half()
        res = torch.randn(M, H, device=dev).half()
        t_re = graph_time(lambda: ops.reorder_fp16_i4(x, idx, quant_mode="sim", clip=0.9))
        t_id = graph_time(lambda: ops.reorder_fp16_i4(x, None, quant_mode="sim", clip=0.9))
        t_rm = graph_time(lambda: ops.rmsnorm_fp16_i4(x, w, idx, 1e-5, quant_mode="sim", clip=0.9))
        t_si = graph_time(lambda: ops.activate_fp16_i4(x, b, quant_mode="sim", clip=0.9))
        print(f"H={H:5d} M={M:3d}: reorder {t_re:5.2f}  (identity order {t_id:5.2f})  rmsnorm {t_rm:5.2f}  silu_mul {t_si:5.2f} us", flush=True)
