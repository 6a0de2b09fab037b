"""s_memtime trace of the quantiser-in-front dot-product kernel (csrc/gemvq_w4a4.hip, tools build): where a launch spends its time.
    ATOM_LIB=build/tools/libatom_hip.so python tools/r06/gemvq_trace.py
Stamps of the first and the last workgroup, per wave: 0 entry, 1 requests issued, 2 quantiser inputs in LDS, 3 behind barrier 1, 4 sum of
squares done, 5 codes written, 6 operand published, 7..10 feature steps 0..3 done, 11 wave done; [14], [15] s_memrealtime (100 MHz)."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from atom_amd import _lib as L  # noqa: E402

L.LIB_PATH = os.path.abspath(os.environ.get("ATOM_LIB", os.path.join(ROOT, "build/tools/libatom_hip.so")))
from atom_amd import ops  # noqa: E402
from tests.helpers import rand_gemm_operands, to_device  # noqa: E402

dev = torch.device("cuda")
NAMES = ["entry", "issued", "inputs->LDS", "barrier1", "sumsq", "codes", "published", "step0", "step1", "step2", "step3", "done", "1/sqrt", "codes pass 0"]


def run(op, N, nseg, K, copies):
    mods = []
    sets = []
    for c in range(copies):
        devs = [to_device(rand_gemm_operands(1, N, K, seed=7 + i), "ref") for i in range(nseg)]
        ms = [types.SimpleNamespace(weight_int4=torch.nn.Parameter(dv[1].clone(), requires_grad=False), weight_int8=torch.nn.Parameter(dv[5].clone(), requires_grad=False),
                                    scale_int4=torch.nn.Parameter(dv[3].clone(), requires_grad=False), scale_int8=torch.nn.Parameter(dv[7].clone(), requires_grad=False)) for dv in devs]
        for md in ms:
            md.packed = (lambda md=md: (md.weight_int4.data, md.weight_int8.data, md.scale_int4.data, md.scale_int8.data))
        sets.append(ops.fuse_projection_weights(ms))
    x = (torch.randn((1, K), device=dev) * 1.3).half()
    x2 = (torch.randn((1, K), device=dev)).half()
    w = (1 + 0.1 * torch.randn(K, device=dev)).half()
    idx = torch.randperm(K, device=dev).to(torch.int16)
    res = torch.randn((1, K), device=dev).half()
    kw = {"reorder": dict(reorder_index=idx), "rmsnorm": dict(x2=w, reorder_index=idx, eps=1e-5),
          "add_rmsnorm": dict(x2=w, residual=res, reorder_index=idx, eps=1e-5), "silu_mul": dict(x2=x2), "merge": None}[op]
    buf = torch.zeros(2 * 16 * 16, dtype=torch.int32, device=dev)
    if op == "merge":                                            # the KV-split merge in front of the reorder quantiser: TRACE_MERGE partial states per head (4 = a layer at context 1024)
        ns = int(os.environ.get("TRACE_MERGE", "4"))
        part = torch.randn((1, K // 128, ns, 130), device=dev)
        part[..., 129] = part[..., 129].abs() + 0.5
        call = lambda st: ops.dense_layer_gemm_i4_merge_q(part, ns, st, reorder_index=idx)
    else:
        call = lambda st: ops.dense_layer_gemm_i4_multi_q(op, x, st, **kw)
    for i in range(3 * copies):
        call(sets[i % copies])
    torch.cuda.synchronize()
    os.environ["ATOM_TRACE_PTR"] = hex(buf.data_ptr())
    call(sets[0])                                                # its weights were evicted by the other copies: a cold launch
    torch.cuda.synchronize()
    del os.environ["ATOM_TRACE_PTR"]
    t = buf.cpu().numpy().astype("uint32").reshape(2, 16, 16)
    print(f"== {op} N={N * nseg} K={K}  ({copies} weight copies)")
    for wg in range(2):
        rt = [(int(t[wg, w, 14]) - int(t[wg, w, 15])) & 0xFFFFFFFF for w in range(16)]
        print(f" workgroup {'first' if wg == 0 else 'last'}: wave lifetime by s_memrealtime {min(rt) / 100:.2f} .. {max(rt) / 100:.2f} us")
        for w in (0, 3, 4, 7, 15):
            base = int(t[wg, w, 0])
            d = [((int(t[wg, w, k]) - base) & 0xFFFFFFFF) for k in range(14)]
            order = [1, 2, 3, 4, 12, 13, 5, 6, 7, 8, 9, 10, 11]
            print(f"  wave {w:2d} cycles since entry: " + "  ".join(f"{NAMES[k]} {d[k]}" for k in order if t[wg, w, k] != 0))


run("rmsnorm", 4096, 3, 4096, 12)
run("reorder", 4096, 1, 4096, 36)
run("add_rmsnorm", 11008, 2, 4096, 8)
run("silu_mul", 4096, 1, 11008, 14)
if os.environ.get("TRACE_MERGE"):
    run("merge", 4096, 1, 4096, 36)
