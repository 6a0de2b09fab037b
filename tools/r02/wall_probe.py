"""Where does the wall-clock overhead of a 20-step timed region go?  (bench.py contract: barrier + synchronize on both sides)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from atom_amd import _lib as L
dev = torch.device("cuda", 0)
lib = L.lib()
M = N = K = 4096
ops_ = bench.make_operands(M, N, K, dev)
a6, b6 = bench.build_f6_operands(ops_, M, N, K, dev)
D = torch.empty((M, N), dtype=torch.float16, device=dev)
ptrs = [a6.data_ptr(), b6.data_ptr()] + [t.data_ptr() for t in ops_[2:]]
stream = torch.cuda.current_stream(dev).cuda_stream
layout = L.SCALE_LAYOUT_PLAIN | L.AB_F6 | L.B_F6S
def step():
    lib.atom_gemm_w4a4_f16(*ptrs, D.data_ptr(), M, N, K, 128, 128, layout, stream)
for _ in range(1500): step()
torch.cuda.synchronize()
def run(mode, steps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if mode != "lazy":
        e0.record(); e1.record()
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps): step()
    t1 = time.perf_counter()
    e1.record()
    if mode == "spin":
        while not e1.query(): pass
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    return (t3 - t0) / steps * 1e6, e0.elapsed_time(e1) / steps * 1e3, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6
import torch.cuda as tc
def run2(kind, steps=20):
    """bench.py's exact sequence"""
    sync = (lambda: tc.synchronize(dev)) if kind == "dev" else (lambda: tc.synchronize())
    for _ in range(5): step()
    sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps): step()
    t1 = time.perf_counter()
    e1.record()
    sync()
    t3 = time.perf_counter()
    return (t3 - t0) / steps * 1e6, e0.elapsed_time(e1) / steps * 1e3, (t1 - t0) * 1e6, 0, (t3 - t1) * 1e6
for kind in ["dev", "nodev", "dev", "nodev"]:
    r = [run2(kind) for _ in range(3)]
    print("bench-seq", kind, " | ".join("wall/step %.1f ev/step %.1f enqueue %.0f wait %.0f sync %.0f" % x for x in r))
for mode in ["lazy", "pre"]:
    r = [run(mode) for _ in range(5)]
    print(mode, " | ".join("wall/step %.1f ev/step %.1f enqueue %.0f wait %.0f sync %.0f" % x for x in r[0:3]))
