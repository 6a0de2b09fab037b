"""Repeatability of the quantiser-in-front dot-product launches (csrc/gemvq_w4a4.hip: LDS-counter sync between quantiser and streamer waves):
3,000 back-to-back launches per op, outputs compared bit for bit with the first one.   python tools/r06/gemvq_stress.py  (on the GPU box)"""
import sys, types, torch
sys.path.insert(0, ".")
from atom_amd import ops
from tests.helpers import rand_gemm_operands, to_device
dev = torch.device("cuda")
def mk(N, K, nseg):
    devs = [to_device(rand_gemm_operands(1, N, K, seed=7 + i), "ref") for i in range(nseg)]
    ms = [types.SimpleNamespace(weight_int4=torch.nn.Parameter(dv[1].clone(), requires_grad=False), weight_int8=torch.nn.Parameter(dv[5].clone(), requires_grad=False),
                                scale_int4=torch.nn.Parameter(dv[3].clone(), requires_grad=False), scale_int8=torch.nn.Parameter(dv[7].clone(), requires_grad=False)) for dv in devs]
    for md in ms:
        md.packed = (lambda md=md: (md.weight_int4.data, md.weight_int8.data, md.scale_int4.data, md.scale_int8.data))
    return ops.fuse_projection_weights(ms)
for op, N, nseg, K in (("rmsnorm", 4096, 3, 4096), ("add_rmsnorm", 11008, 2, 4096), ("reorder", 4096, 1, 4096), ("silu_mul", 4096, 1, 11008)):
    fused = mk(N, K, nseg)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.randn((1, K), device=dev, generator=g) * 1.3).half()
    x2 = torch.randn((1, K), device=dev, generator=g).half()
    w = (1 + 0.1 * torch.randn(K, device=dev, generator=g)).half()
    idx = torch.randperm(K, device=dev).to(torch.int16)
    res = torch.randn((1, K), device=dev, generator=g).half()
    kw = {"reorder": dict(reorder_index=idx), "rmsnorm": dict(x2=w, reorder_index=idx, eps=1e-5),
          "add_rmsnorm": dict(x2=w, residual=res, reorder_index=idx, eps=1e-5), "silu_mul": dict(x2=x2)}[op]
    want, _ = ops.dense_layer_gemm_i4_multi_q(op, x, fused, **kw)
    want = [t.clone() for t in want]
    bad = 0
    for it in range(3000):
        got, _ = ops.dense_layer_gemm_i4_multi_q(op, x, fused, **kw)
        if it % 50 == 0 or it > 2950:
            bad += int(not all(torch.equal(a, b) for a, b in zip(got, want)))
    torch.cuda.synchronize()
    got, _ = ops.dense_layer_gemm_i4_multi_q(op, x, fused, **kw)
    bad += int(not all(torch.equal(a, b) for a, b in zip(got, want)))
    print(op, N * nseg, K, "mismatches", bad)

# the KV-split merge in front of the reorder quantiser (q_op 5: roles + the (m, d) exchange through wave-local LDS), 8 and 16 states per head
for splits in (8, 16, 3):
    fused = mk(4096, 4096, 1)
    g = torch.Generator(device="cuda").manual_seed(splits)
    part = torch.randn((1, 32, splits, 130), device=dev, generator=g)
    part[..., 128] *= 4
    part[..., 129] = part[..., 129].abs() + 0.25
    idx = torch.randperm(4096, device=dev).to(torch.int16)
    (want,) = ops.dense_layer_gemm_i4_merge_q(part, splits, fused, reorder_index=idx)
    want = want.clone()
    bad = 0
    for it in range(3000):
        (got,) = ops.dense_layer_gemm_i4_merge_q(part, splits, fused, reorder_index=idx)
        if it % 50 == 0 or it > 2950:
            bad += int(not torch.equal(got, want))
    torch.cuda.synchronize()
    print("merge", splits, "states: mismatches", bad)
