"""INT4 paged KV-cache bookkeeping with the API of the reference's ``punica/utils/kvcache.py`` (KvPoolInt4 :6-55,
KvCacheInt4 :58-98, BatchedKvCacheInt4 :101-128): same class / property / method names and the same tensor layouts, so
code written against the reference's objects runs on these.  Host-side only; the device work is in atom_amd.ops
(init_kv_i4 / append_kv_i4 / batch_decode_i4 -> csrc/kv_i4.hip).

Pool layout (reference kvcache.py:17-26, page.cuh:78-110):
    buf    uint8 [capacity, num_layers, 2, num_heads, block_len, head_dim // 2]   packed u4, K at [.., 0, ..], V at [.., 1, ..]
    param  fp16  [capacity, num_layers, 2, num_heads, block_len, 2]              (scale, zero) per token and head
"""
from __future__ import annotations

from typing import Sequence

import torch


class KvPoolInt4:
    def __init__(self, num_layers: int, num_heads: int, head_dim: int, capacity: int, block_len: int,
                 device: torch.device):
        shape = (capacity, num_layers, 2, num_heads, block_len)
        self._buf = torch.empty(shape + (head_dim // 2,), dtype=torch.uint8, device=device)
        self._param = torch.empty(shape + (2,), dtype=torch.float16, device=device)
        self._free = set(range(capacity))

    buf = property(lambda self: self._buf)
    param = property(lambda self: self._param)
    num_layers = property(lambda self: self._buf.shape[1])
    block_len = property(lambda self: self._buf.shape[4])
    num_free_blocks = property(lambda self: len(self._free))

    def alloc_block(self) -> int:
        return self._free.pop()

    def free_block(self, idx: int):
        if not 0 <= idx < self._buf.size(0) or idx in self._free:
            raise AssertionError(f"block {idx} is not an allocated block of this pool")
        self._free.add(idx)


class KvCacheInt4:
    """Pages of ONE sequence."""

    def __init__(self, pool: KvPoolInt4, init_len: int):
        if init_len < 0:
            raise ValueError("init_len must be non-negative")
        self._pool = pool
        self._seqlen = init_len
        self._indicies = [pool.alloc_block() for _ in range(-(-init_len // pool.block_len))]

    pool = property(lambda self: self._pool)
    seqlen = property(lambda self: self._seqlen)
    indicies = property(lambda self: self._indicies)          # (sic) the reference's spelling

    def acquire_one(self):
        """Make room for one more token (a new page when the last one is full)."""
        if self._seqlen % self._pool.block_len == 0 and len(self._indicies) * self._pool.block_len == self._seqlen:
            self._indicies.append(self._pool.alloc_block())
        self._seqlen += 1

    def release(self):
        for idx in self._indicies:
            self._pool.free_block(idx)
        self._indicies.clear()
        self._seqlen = 0


class BatchedKvCacheInt4:
    """Device-side page tables of a batch of sequences: what the kernels take."""

    def __init__(self, kv: Sequence[KvCacheInt4]):
        assert len(kv) > 0
        pool = kv[0].pool
        assert all(c.pool is pool for c in kv)
        device = pool.buf.device
        counts = [len(c.indicies) for c in kv]
        self.data = pool.buf
        self.param = pool.param
        self.indptr = torch.tensor([0] + list(torch.tensor(counts).cumsum(0).tolist()), dtype=torch.int32, device=device)
        self.indicies = torch.tensor([i for c in kv for i in c.indicies], dtype=torch.int32, device=device)
        self.last_page_offset = torch.tensor([(c.seqlen - 1) % pool.block_len + 1 for c in kv], dtype=torch.int32,
                                             device=device)
        self.max_pages = max(counts)            # host-side hint for the decode kernel's KV split (not in the reference)

    @property
    def page_size(self):
        return self.data.size(-2)
