"""gate_proj + up_proj + SiLU x up + quant: three launches against the fused one, Llama-7B sizes.  us per MLP front half."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from atom_amd import ops

def t(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

g = torch.Generator(device="cuda").manual_seed(0)
N, K = 11008, 4096
W = lambda: ops.quant_weight_w4((torch.randn((N, K), device="cuda", generator=g) * 0.03).half(), 0.85, 2)
gate, up = W(), W()
fused = ops.fuse_gate_up_weights(gate, up)
bg, bu = ops.repack_weight_f6(gate[0], gate[2]), ops.repack_weight_f6(up[0], up[2])
print("# M | three launches (gate GEMM + up GEMM + silu_mul_quant, F6 operands) | fused | saved")
for M in (512, 1024, 2048, 4096, 16384, 65536):
    x = torch.randn((M, K), device="cuda", generator=g).half()
    a = ops.reorder_fp16_i4(x, None, quant_mode="sim", clip=0.9, scale_layout="plain", wide_codes="f6")
    def three():
        yg = ops.dense_layer_gemm_i4_fp16(a[1], bg, a[3], gate[2], a[0], gate[1], a[2], gate[3], scale_layout="plain", a_wide="f6")
        yu = ops.dense_layer_gemm_i4_fp16(a[1], bu, a[3], up[2], a[0], up[1], a[2], up[3], scale_layout="plain", a_wide="f6")
        return ops.activate_fp16_i4(yg, yu, quant_mode="sim", clip=0.9, scale_layout="plain", wide_codes="f6")
    def one():
        return ops.gate_up_silu_quant_f6(a[1], a[0], a[2], fused, quant_mode="sim", clip=0.9, scale_layout="plain")
    it = 50 if M <= 4096 else 5
    t3, t1 = t(three, it), t(one, it)
    print(f"{M:6d} | {t3:10.1f} us | {t1:10.1f} us | {t3 - t1:8.1f} us ({(1 - t1 / t3) * 100:.1f} %)", flush=True)
