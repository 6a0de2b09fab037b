"""INT4 paged-KV decode attention as a HIP graph (GPU time without the Python dispatch): python tools/decode_graph_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
import os  # noqa: E402
from atom_amd import _lib as L  # noqa: E402
if os.environ.get("ATOM_LIB"):                            # a tuning build of the library (ATOM_* overrides are read by that build only)
    L.LIB_PATH = os.path.abspath(os.environ["ATOM_LIB"])
from atom_amd import ops  # noqa: E402
from atom_amd.utils.kvcache import BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4  # noqa: E402


def run(batch, seqlen, heads=32, block=16, reps=20):
    dev = torch.device("cuda")
    pool = KvPoolInt4(1, heads, 128, batch * (-(-seqlen // block)), block, dev)
    pool.buf.copy_(torch.randint(0, 256, pool.buf.shape, device=dev, dtype=torch.uint8))
    pool.param.copy_((torch.rand(pool.param.shape, device=dev) * 0.2 + 0.01).half())
    kv = BatchedKvCacheInt4([KvCacheInt4(pool, seqlen) for _ in range(batch)])
    q = torch.randn((batch, heads, 128), device=dev).half()
    side = torch.cuda.Stream()                            # the split workspace is per (device, stream): warm up on the capture stream
    with torch.cuda.stream(side):
        for _ in range(3):
            ops.batch_decode_i4(q, kv, 0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            ops.batch_decode_i4(q, kv, 0)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (5 * reps) * 1e3
    mb = batch * heads * seqlen * 2 * (64 + 4) / 1e6
    print(f"batch {batch:4d} ctx {seqlen:5d}: {us:7.1f} us  {mb / us * 1e-0:6.2f} GB/s x1000 ({mb:.0f} MB)")


if __name__ == "__main__":
    for b, s in [(1, 1024), (4, 1024), (16, 1024), (64, 1024), (1, 4096), (16, 4096), (128, 2048)]:
        run(b, s)
