"""Bench of the INT4 paged-KV decode attention (HBM-bound): python tools/decode_bench.py  (on the GPU box)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from atom_amd import ops  # noqa: E402
from atom_amd.utils.kvcache import BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4  # noqa: E402


def run(batch, seqlen, heads=32, layers=1, block=16, iters=50):
    dev = torch.device("cuda")
    cap = batch * (-(-seqlen // block))
    pool = KvPoolInt4(layers, heads, 128, cap, block, dev)
    pool.buf.copy_(torch.randint(0, 256, pool.buf.shape, device=dev, dtype=torch.uint8))
    pool.param.copy_((torch.rand(pool.param.shape, device=dev) * 0.2 + 0.01).half())
    kv = BatchedKvCacheInt4([KvCacheInt4(pool, seqlen) for _ in range(batch)])
    q = torch.randn((batch, heads, 128), device=dev).half()
    for _ in range(5):
        ops.batch_decode_i4(q, kv, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.batch_decode_i4(q, kv, 0)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    bytes_ = batch * heads * seqlen * 2 * (64 + 4) + 2 * batch * heads * 128 * 2
    print(f"DRESULT batch={batch} seqlen={seqlen} heads={heads} page={block}  {us:8.1f} us  {bytes_ / us / 1e3:7.0f} GB/s "
          f"algorithmic ({bytes_ / 1e6:.1f} MB)", flush=True)


if __name__ == "__main__":
    for b, s in [(1, 2048), (8, 2048), (32, 2048), (32, 512), (128, 1024), (4, 16384), (64, 4096)]:
        run(b, s)
