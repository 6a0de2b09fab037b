"""Decode GEMV (M = 1) under HIP-graph replay: staged kernel vs the one-round-trip kernel, and the floor of an empty launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes
import bench
dev = torch.device("cuda", 0)
def graph_time(make_fn, iters=100):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn = make_fn(torch.cuda.current_stream().cuda_stream)
        for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fn = make_fn(torch.cuda.current_stream().cuda_stream)
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
libpath = os.environ.get("ATOM_LIB", os.path.join(os.path.dirname(bench.__file__), "atom_amd", "libatom_hip.so"))
lib = ctypes.CDLL(libpath)
lib.atom_gemm_w4a4_f16.argtypes = [ctypes.c_void_p] * 9 + [ctypes.c_int64] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
x = torch.zeros(64, device=dev)
print("tiny torch kernel (x.add_(1)) under graph replay: %.2f us" % graph_time(lambda st: (lambda: x.add_(1))))
for (N, K) in ((4096, 4096), (5120, 5120), (13824, 5120), (5120, 13824), (11008, 4096), (4096, 11008)):
    ops_ = bench.make_operands(1, N, K, dev, seed=1)
    D = torch.empty((1, N), dtype=torch.float16, device=dev)
    ptrs = [t.data_ptr() for t in ops_]
    t = graph_time(lambda st: (lambda: lib.atom_gemm_w4a4_f16(*ptrs, D.data_ptr(), 1, N, K, 128, 128, 1, st)))
    by = bench.algorithmic_bytes(1, N, K)
    print(f"M=1 N={N:6d} K={K:6d}: {t:6.2f} us  {by / t / 1e6:5.2f} TB/s = {by / t / 1e6 / 8:.3f} of 8 TB/s", flush=True)
