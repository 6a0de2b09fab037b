"""Packed operands through atom_gemm_w4a4_f16_ws at mid-size M: time per call (rocprofv3 --kernel-trace --stats around this script gives
the split between the re-coding launch and the GEMM)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "r02"))
import torch
import bench
from atom_amd import _lib as L
from sweep import time_call
dev = torch.device("cuda", 0)
lib = L.lib()
stream = torch.cuda.current_stream(dev).cuda_stream
for M in [int(x) for x in (sys.argv[1:] or ["512", "768", "1024", "1536"])]:
    N = K = 4096
    ops_ = bench.make_operands(M, N, K, dev, seed=1)
    D = torch.empty((M, N), dtype=torch.float16, device=dev)
    ptrs = [t.data_ptr() for t in ops_]
    wsb = lib.atom_gemm_w4a4_workspace_bytes(M, N, K)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    t_ws = time_call(lambda: lib.atom_gemm_w4a4_f16_ws(*ptrs, D.data_ptr(), M, N, K, 128, 128, L.SCALE_LAYOUT_PLAIN, ws.data_ptr(), wsb, stream), 100)
    t_plain = time_call(lambda: lib.atom_gemm_w4a4_f16(*ptrs, D.data_ptr(), M, N, K, 128, 128, L.SCALE_LAYOUT_PLAIN, stream), 100)
    print(f"M={M}: workspace route {t_ws:.2f} us (workspace {wsb} B), plain entry {t_plain:.2f} us", flush=True)
