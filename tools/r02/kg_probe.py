"""F6 tile kernels with in-workgroup K groups (cfg 5 / 6: 128x128, two groups, 2 / 3 stages; cfg 9: 64x128, two groups; cfg 12: 64x128, four) against the 4-wave 128x128
kernel (cfg 3) and cfg 3 + split-K 2; "default" = what the dispatch picks.  Time and bits.
Needs the tools build (make -C atom_amd/csrc tools): the geometry is forced through ATOM_F6_CFG / ATOM_F6_SPLITS3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from atom_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "build", "tools", "libatom_hip.so")
import bench
sys.path.insert(0, os.path.join(ROOT, "tools", "r02"))
from sweep import time_call

dev = torch.device("cuda", 0)
lib = L.lib()
stream = torch.cuda.current_stream(dev).cuda_stream


def run(M, N, K):
    ops_ = bench.make_operands(M, N, K, dev, seed=1)
    ptrs = [t.data_ptr() for t in ops_]
    a6, b6 = bench.build_f6_operands(ops_, M, N, K, dev)
    p6 = [a6.data_ptr(), b6.data_ptr()] + ptrs[2:]
    wsb = max(lib.atom_gemm_w4a4_workspace_bytes(M, N, K), 2 * M * N * 4)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    fl = L.SCALE_LAYOUT_PLAIN | L.AB_F6 | L.B_F6S
    out = {}
    res = {}
    for name, env in (("cfg3", {"ATOM_F6_CFG": "3", "ATOM_F6_SPLITS3": "1"}), ("cfg3s2", {"ATOM_F6_CFG": "3", "ATOM_F6_SPLITS3": "2"}),
                      ("cfg5", {"ATOM_F6_CFG": "5"}), ("cfg6", {"ATOM_F6_CFG": "6"}), ("cfg7", {"ATOM_F6_CFG": "9"}), ("cfg8", {"ATOM_F6_CFG": "12"}), ("cfg59", {})):
        for k in ("ATOM_F6_CFG", "ATOM_F6_SPLITS3"):
            os.environ.pop(k, None)
        os.environ.update(env)
        D = torch.zeros((M, N), dtype=torch.float16, device=dev)
        fn = lambda: lib.atom_gemm_w4a4_f16_ws(*p6, D.data_ptr(), M, N, K, 128, 128, fl, ws.data_ptr(), wsb, stream)
        st = fn()
        assert st == 0, (name, st)
        torch.cuda.synchronize()
        out[name] = D.clone()
        res[name] = time_call(fn, 100)
    same = torch.equal(out["cfg6"], out["cfg5"]) and torch.equal(out["cfg7"], out["cfg5"])
    close = (out["cfg8"].float() - out["cfg3"].float()).abs().max().item()
    print(f"{M:5d}x{N:5d}x{K:5d}  cfg3 {res['cfg3']:7.2f}  cfg3+s2 {res['cfg3s2']:7.2f}  cfg5 {res['cfg5']:7.2f}  cfg6 {res['cfg6']:7.2f}  cfg9 {res['cfg7']:7.2f}  cfg12 {res['cfg8']:7.2f}  default {res['cfg59']:7.2f} us   "
          f"consistent {same}  max|cfg12-cfg3| {close:.4f}", flush=True)


for M in (128, 256, 384, 512, 768, 1024, 1536, 2048):
    run(M, 4096, 4096)
for (N, K) in ((5120, 5120), (13824, 5120), (5120, 13824), (11008, 4096), (4096, 11008)):
    for M in (256, 512, 1024):
        run(M, N, K)
