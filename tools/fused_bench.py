"""Residual add + RMSNorm-quant: separate (torch add, then atom_rmsnorm_reorder_quant_f16) vs fused.  Run on the GPU box."""
import sys

import torch

sys.path.insert(0, ".")
from atom_amd import ops  # noqa: E402


def t(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M in (4096, 65536):
    H = 4096
    x = torch.randn(M, H, device="cuda").half()
    res = torch.randn(M, H, device="cuda").half()
    w = torch.ones(H, device="cuda").half()
    idx = torch.randperm(H, device="cuda").to(torch.int16)
    a = t(lambda: ops.rmsnorm_fp16_i4(x + res, w, idx, 1e-5))
    b = t(lambda: ops.add_rmsnorm_fp16_i4(x, res, w, idx, 1e-5))
    print(f"FRESULT M={M} H={H}: add + rmsnorm-quant {a:.1f} us, fused {b:.1f} us ({a / b:.2f}x)", flush=True)
