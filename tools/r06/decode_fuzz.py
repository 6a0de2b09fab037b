"""Randomised check of the decode attention after round 6's rebuild (csrc/kv_i4.hip): for random batches / heads / page sizes / lengths,
(a) against the numpy oracle (2e-3 of the output scale), (b) the fused append against append-then-decode (bit for bit, cache bytes too),
(c) split against unsplit KV ranges (2e-3).   python tools/r06/decode_fuzz.py [cases]   (on the GPU box)
The last line carries a SHA-256 over every output's bits: two builds (ATOM_LIB=...) that print the same digest are bit-identical on these cases."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from atom_amd import ops  # noqa: E402
from atom_amd.utils.kvcache import BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4  # noqa: E402
from oracle import atom_oracle as O  # noqa: E402  (the checker)

dev = torch.device("cuda")
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t2n = lambda t: t.detach().cpu().numpy()
bad = 0
digest = hashlib.sha256()
for case in range(ncases):
    heads = int(rng.choice([1, 4, 8, 32]))
    block = int(rng.choice([16, 32, 64]))
    B = int(rng.choice([1, 2, 3, 8, 9, 16, 33, 70]))
    if B * heads > 1200:
        B = max(1, 1200 // heads)
    hi = int(rng.choice([40, 300, 1100, 2100]))
    seqlens = [int(x) for x in rng.integers(1, hi + 1, size=B)]
    cap = sum(-(-n // block) for n in seqlens) + 2
    outs, pools = [], []
    g = torch.Generator(device="cuda").manual_seed(case)
    k32 = torch.randn((B, heads * 128), device=dev, generator=g) * 2
    v32 = torch.randn((B, heads * 128), device=dev, generator=g) * 3 + 0.5
    q = torch.randn((B, heads, 128), device=dev, generator=g).half()
    for fused in (False, True):
        pool = KvPoolInt4(num_layers=2, num_heads=heads, head_dim=128, capacity=cap, block_len=block, device=dev)
        gp = torch.Generator(device="cuda").manual_seed(11)
        pool.buf.copy_(torch.randint(0, 256, pool.buf.shape, device=dev, dtype=torch.uint8, generator=gp))
        pool.param.copy_((torch.rand(pool.param.shape, device=dev, generator=gp) * 0.05 + 0.01).half())
        pool._free = set(range(cap))
        kv = BatchedKvCacheInt4([KvCacheInt4(pool, n) for n in seqlens])
        if fused:
            o = ops.batch_decode_i4(q, kv, 1, append_kv=(k32, v32))
        else:
            ops.quant_append_kv_i4(kv, k32, v32, 1)
            o = ops.batch_decode_i4(q, kv, 1)
        outs.append(o)
        pools.append((pool, kv))
    ok_b = torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)) and torch.equal(pools[0][0].buf, pools[1][0].buf)
    pool, kv = pools[0]
    splits = ops.decode_splits(B, kv)
    kv.max_pages = 0
    o1 = ops.batch_decode_i4(q, kv, 1)
    for o_ in (outs[0], o1):
        digest.update(t2n(o_.view(torch.int16)).tobytes())
    ok_c = torch.allclose(o1.float(), outs[0].float(), atol=2e-3 * float(o1.float().abs().max()) + 1e-3, rtol=0)
    ok_a = True
    if sum(seqlens) * heads <= 40000:                       # the numpy oracle is slow
        tables = (t2n(kv.indptr), t2n(kv.indicies), t2n(kv.last_page_offset))
        ref = O.batch_decode_i4(t2n(q), t2n(pool.buf), t2n(pool.param), *tables, 1)
        ok_a = np.abs(t2n(o1).astype(np.float64) - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-3
    if not (ok_a and ok_b and ok_c):
        bad += 1
        print(f"case {case}: heads {heads} block {block} B {B} max len {max(seqlens)} splits {splits}: oracle {ok_a} fused {ok_b} split {ok_c}")
print(f"{ncases} cases, {bad} bad; outputs sha256 {digest.hexdigest()[:16]}")
