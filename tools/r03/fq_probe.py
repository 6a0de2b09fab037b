"""Stand-alone timing of atom_gemm_w4a4_multi_q against the quantiser op + atom_gemm_w4a4_multi (HIP-graph replay, us per pair)."""
import sys, types
import torch
sys.path.insert(0, ".")
from atom_amd import ops
from tests.helpers import rand_gemm_operands, to_device


def graph_us(fn, n=40):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for (op, M, N, K, nseg) in [("reorder", 1, 4096, 4096, 1), ("rmsnorm", 1, 4096, 4096, 3), ("add_rmsnorm", 1, 11008, 4096, 2), ("silu_mul", 1, 4096, 11008, 1),
                            ("rmsnorm", 2, 4096, 4096, 3), ("silu_mul", 2, 4096, 11008, 1)]:
    devs = [to_device(rand_gemm_operands(4, N, K, seed=51 + i), "ref") for i in range(nseg)]
    mods = [types.SimpleNamespace(weight_int4=torch.nn.Parameter(dv[1], requires_grad=False), weight_int8=torch.nn.Parameter(dv[5], requires_grad=False),
                                  scale_int4=torch.nn.Parameter(dv[3], requires_grad=False), scale_int8=torch.nn.Parameter(dv[7], requires_grad=False)) for dv in devs]
    for md in mods:
        md.packed = (lambda md=md: (md.weight_int4.data, md.weight_int8.data, md.scale_int4.data, md.scale_int8.data))
    fused = ops.fuse_projection_weights(mods)
    x = (torch.randn((M, K), device="cuda") * 1.3).half(); x2 = (torch.randn((M, K), device="cuda") * 0.8).half()
    w = (1 + 0.1 * torch.randn(K, device="cuda")).half(); idx = torch.randperm(K, device="cuda").to(torch.int16)
    res = (torch.randn((M, K), device="cuda") * 2).half()
    if op == "reorder":
        q = lambda: ops.reorder_fp16_i4(x, idx); kw = dict(reorder_index=idx)
    elif op == "rmsnorm":
        q = lambda: ops.rmsnorm_fp16_i4(x, w, idx, 1e-5); kw = dict(x2=w, reorder_index=idx, eps=1e-5)
    elif op == "add_rmsnorm":
        q = lambda: ops.add_rmsnorm_fp16_i4(x, res, w, idx, 1e-5)[1:]; kw = dict(x2=w, residual=res, reorder_index=idx, eps=1e-5)
    else:
        q = lambda: ops.activate_fp16_i4(x, x2); kw = dict(x2=x2)

    def sep():
        o, n, os_, ns = q()
        ops.dense_layer_gemm_i4_multi(n, ns, o, os_, fused)
    qt = q()
    t_q = graph_us(lambda: q())
    t_g = graph_us(lambda: ops.dense_layer_gemm_i4_multi(qt[1], qt[3], qt[0], qt[2], fused))
    t_sep = graph_us(sep)
    t_f = graph_us(lambda: ops.dense_layer_gemm_i4_multi_q(op, x, fused, **kw))
    print(f"{op:12s} M={M} N={N}x{nseg} K={K}: quantiser {t_q:5.2f}  gemm {t_g:5.2f}  both {t_sep:5.2f}  fused {t_f:5.2f} us", flush=True)
