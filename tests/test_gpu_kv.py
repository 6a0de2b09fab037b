"""GPU tests of the INT4 paged-KV ops (atom_kv_append_i4, atom_batch_decode_i4) against the numpy restatement of the
reference CUDA kernels (oracle.kv_append_i4 / batch_decode_i4) AND against the reference's own CPU implementations of the same
two operations (cpu_reference::append_paged_kv_cache / single_quantize_mha, kernels/src/flashinfer/cpu_reference.h, compiled from
/root/reference into oracle/_ref/libatom_ref.so; tests/test_oracle_ref.py pins the restatement against them on the CPU)."""
import numpy as np
import pytest
import torch

from oracle import atom_oracle as O
from tests.helpers import t2n

pytestmark = pytest.mark.gpu


def _setup(seqlens, layers=2, heads=4, block=16, seed=0, extra_blocks=3):
    from atom_amd.utils.kvcache import BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4
    g = torch.Generator(device="cuda").manual_seed(seed)
    cap = sum(-(-s // block) for s in seqlens) + extra_blocks
    pool = KvPoolInt4(layers, heads, 128, cap, block, torch.device("cuda"))
    pool.buf.copy_(torch.randint(0, 256, pool.buf.shape, device="cuda", generator=g, dtype=torch.uint8))
    pool.param.copy_((torch.rand(pool.param.shape, device="cuda", generator=g) * 0.2 + 0.01).half())
    cs = [KvCacheInt4(pool, s) for s in seqlens]
    return pool, cs, BatchedKvCacheInt4(cs), g


def _np_tables(kv):
    return t2n(kv.indptr), t2n(kv.indicies), t2n(kv.last_page_offset)


@pytest.mark.parametrize("seqlens,heads,block", [([37, 5, 16], 4, 16), ([1], 32, 16), ([100, 33, 64, 2, 17], 8, 32),
                                                  ([500, 3], 32, 16)])
def test_init_and_append_kv_bit_exact(seqlens, heads, block):
    from atom_amd import ops
    pool, cs, kv, g = _setup(seqlens, heads=heads, block=block)
    T = sum(seqlens)
    k = torch.randint(0, 256, (T, heads, 64), device="cuda", generator=g, dtype=torch.uint8)
    v = torch.randint(0, 256, (T, heads, 64), device="cuda", generator=g, dtype=torch.uint8)
    kp = (torch.rand((T, heads, 2), device="cuda", generator=g) + 0.5).half()
    vp = (torch.rand((T, heads, 2), device="cuda", generator=g) + 0.5).half()
    ind = torch.tensor(np.cumsum([0] + seqlens), dtype=torch.int32, device="cuda")
    data, param = t2n(pool.buf).copy(), t2n(pool.param).copy()
    ops.init_kv_i4(kv, k, v, kp, vp, ind, 1)
    O.kv_append_i4(data, param, *_np_tables(kv), t2n(k), t2n(v), t2n(kp), t2n(vp), 1, t2n(ind))
    assert np.array_equal(t2n(pool.buf), data) and np.array_equal(t2n(pool.param).view(np.uint16), param.view(np.uint16))
    # decode step: one more token per sequence (page boundaries included: lengths 16 / 64 get a new page)
    from atom_amd.utils.kvcache import BatchedKvCacheInt4
    for c in cs:
        c.acquire_one()
    kv2 = BatchedKvCacheInt4(cs)
    B = len(seqlens)
    k1 = torch.randint(0, 256, (B, heads, 64), device="cuda", generator=g, dtype=torch.uint8)
    v1 = torch.randint(0, 256, (B, heads, 64), device="cuda", generator=g, dtype=torch.uint8)
    kp1 = (torch.rand((B, heads, 2), device="cuda", generator=g) + 0.5).half()
    ops.append_kv_i4(kv2, k1, v1, kp1, kp1, 0)
    O.kv_append_i4(data, param, *_np_tables(kv2), t2n(k1), t2n(v1), t2n(kp1), t2n(kp1), 0, None)
    assert np.array_equal(t2n(pool.buf), data) and np.array_equal(t2n(pool.param).view(np.uint16), param.view(np.uint16))


@pytest.mark.parametrize("seqlens,heads,block", [([37, 5, 16, 1], 4, 16), ([300], 32, 16), ([129, 64, 250], 8, 32),
                                                  ([2000, 7], 4, 16), ([300, 17, 260, 33, 1, 290, 128, 64], 32, 16),
                                                  ([150] * 19 + [3], 32, 16), ([210, 97, 5], 5, 48), ([300] * 7 + [49], 40, 48)])
def test_batch_decode_matches_oracle(seqlens, heads, block):
    """The last two: head counts and page lengths that are not powers of two (40 heads = Llama-13B; 3 tiles per page) -- the kernel divides
    by them through host-prepared reciprocals.  Tolerance 2e-3 of the output scale: FP32 accumulation, hardware exp2, incremental RoPE rotation (the reference
    itself uses __sincosf / __powf fast intrinsics, decode.cuh:63-66,537).  The last two shapes have >= 256 (sequence, head) pairs
    and a split KV range: the splits are waves of 12-wave workgroups that merge in LDS (4 splits x 3 pairs, 256 pairs: the last
    workgroup has spare waves; 2 splits x 6 pairs)."""
    from atom_amd import ops
    pool, cs, kv, g = _setup(seqlens, heads=heads, block=block, seed=len(seqlens))
    q = torch.randn((len(seqlens), heads, 128), device="cuda", generator=g).half()
    for layer in (0, 1):
        o = ops.batch_decode_i4(q, kv, layer)
        ref = O.batch_decode_i4(t2n(q), t2n(pool.buf), t2n(pool.param), *_np_tables(kv), layer)
        err = np.abs(t2n(o).astype(np.float64) - ref).max()
        assert err <= 2e-3 * np.abs(ref).max() + 1e-3, (layer, err, np.abs(ref).max())
    # KV split and no split agree (flash-decoding merge)
    o1 = ops.batch_decode_i4(q, kv, 0)
    kv.max_pages = 0
    o2 = ops.batch_decode_i4(q, kv, 0)
    assert torch.allclose(o1.float(), o2.float(), atol=2e-3 * float(o2.float().abs().max()), rtol=0)


def test_batch_decode_properties_and_errors():
    from atom_amd import ops
    from atom_amd._lib import AtomHipError
    pool, cs, kv, g = _setup([70, 33], heads=4)
    # identical keys -> uniform softmax -> output = mean of the de-quantised values (position-independent: all-equal
    # keys are only equal after RoPE if they are zero, so use zero keys: nibble 8, scale 1, zero 8)
    pool.buf[:, :, 0] = 0x88
    pool.param[:, :, 0, :, :, 0] = 1.0
    pool.param[:, :, 0, :, :, 1] = 8.0
    q = torch.randn((2, 4, 128), device="cuda", generator=g).half()
    o = ops.batch_decode_i4(q, kv, 0)
    ref = O.batch_decode_i4(t2n(q), t2n(pool.buf), t2n(pool.param), *_np_tables(kv), 0)
    assert np.abs(t2n(o) - ref).max() <= 2e-3 * np.abs(ref).max()
    # entries past a sequence's end are uninitialised memory in real use: NaN / Inf there must not leak into the output
    from atom_amd.utils.kvcache import BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4
    pool3 = KvPoolInt4(1, 4, 128, 8, 16, torch.device("cuda"))
    pool3.buf.fill_(0xFF)
    pool3.param.fill_(float("nan"))
    c3 = [KvCacheInt4(pool3, 21), KvCacheInt4(pool3, 3)]
    kv3 = BatchedKvCacheInt4(c3)
    T = 24
    ops.init_kv_i4(kv3, torch.randint(0, 256, (T, 4, 64), device="cuda", dtype=torch.uint8),
                   torch.randint(0, 256, (T, 4, 64), device="cuda", dtype=torch.uint8),
                   (torch.rand((T, 4, 2), device="cuda") + 0.1).half(), (torch.rand((T, 4, 2), device="cuda") + 0.1).half(),
                   torch.tensor([0, 21, 24], dtype=torch.int32, device="cuda"), 0)
    o3 = ops.batch_decode_i4(q, kv3, 0)
    assert torch.isfinite(o3).all()
    par = t2n(pool3.param)
    ref3 = O.batch_decode_i4(t2n(q), t2n(pool3.buf), np.nan_to_num(par), *_np_tables(kv3), 0)
    assert np.abs(t2n(o3) - ref3).max() <= 2e-3 * np.abs(ref3).max() + 1e-3
    data = t2n(pool.buf)
    with pytest.raises(AtomHipError):
        ops.batch_decode_i4(q.cpu(), kv, 0)
    with pytest.raises(AtomHipError):
        ops.batch_decode_i4(q, kv, 5)                        # layer out of range
    assert data.shape[1] == 2


def test_kv_fake_quant_bit_exact(golden_dir):
    """atom_kv_fake_quant_f16 against the reference-generated golden and the oracle, contiguous and transposed-view
    inputs (the attention layer passes the transposed projection output, qLlamaLayer.py:238-249)."""
    import os
    from atom_amd import ops
    from tests.helpers import bits16
    z = np.load(os.path.join(golden_dir, "kv_fake_quant.npz"))
    x = torch.from_numpy(z["x"]).cuda()
    for key in z.files:
        if key == "x":
            continue
        bits, clip = int(key.split("_b")[1].split("_")[0]), float(key.split("_c")[1])
        y = ops.kv_fake_quant(x, bits, clip)
        assert np.array_equal(bits16(t2n(y)), bits16(z[key])), key
    xt = x.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    assert not xt.is_contiguous()
    y = ops.kv_fake_quant(xt, 4, 1.0)
    assert y.is_contiguous() and np.array_equal(bits16(t2n(y)), bits16(z["k_b4_c1.0"]))
    g = torch.Generator(device="cuda").manual_seed(1)
    big = (torch.randn((3, 32, 257, 128), device="cuda", generator=g) * 2).half()
    want = O.kv_fake_quant_sim(t2n(big), 4, 0.9)
    assert np.array_equal(bits16(t2n(ops.kv_fake_quant(big, 4, 0.9))), bits16(want))
    # through the drop-in wrapper
    import types
    from atom_amd.model import quant as Q
    args = types.SimpleNamespace(abits=4, kv_clip_ratio=1.0)
    assert np.array_equal(bits16(t2n(Q.quantize_attn_k_wrapper(xt, args))), bits16(z["k_b4_c1.0"]))


@pytest.mark.parametrize("seqlens,heads,block", [([5, 17, 32, 1], 4, 16), ([40] * 9, 2, 32)])
def test_fused_quant_append_equals_o4_gemm_plus_append(seqlens, heads, block):
    """Decode step: atom_gemm_w4a4_f32 + atom_kv_quant_append_f32 (FP32 k / v sums -> u4 codes straight into the cache)
    leaves exactly the cache contents of the reference's op sequence dense_layer_gemm_i4_o4 (decode path) + append_kv_i4,
    and touches nothing but the last token's slots."""
    from atom_amd import ops
    from atom_amd.utils import BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4
    from tests.helpers import rand_gemm_operands, to_device
    dev = torch.device("cuda")
    B, N, K = len(seqlens), heads * 128, 640
    dk = to_device(rand_gemm_operands(B, N, K, seed=3), "ref")
    dv = to_device(rand_gemm_operands(B, N, K, seed=4), "ref")
    assert ops.decode_gemm_fits(B, N, K)
    pools = []
    for fused in (False, True):
        torch.manual_seed(1)
        pool = KvPoolInt4(num_layers=2, num_heads=heads, head_dim=128, capacity=64, block_len=block, device=dev)
        pool.buf.copy_(torch.randint(0, 256, pool.buf.shape, dtype=torch.uint8))
        pool.param.copy_(torch.rand(pool.param.shape).half())
        pool._free = set(range(64))                                  # same page numbering in both runs
        kv = BatchedKvCacheInt4([KvCacheInt4(pool, n) for n in seqlens])
        before = (pool.buf.clone(), pool.param.clone())
        if fused:
            ops.quant_append_kv_i4(kv, ops.dense_layer_gemm_i4_f32(*dk), ops.dense_layer_gemm_i4_f32(*dv), 1)
        else:
            k, ks = ops.dense_layer_gemm_i4_o4(*dk)
            v, vs = ops.dense_layer_gemm_i4_o4(*dv)
            ops.append_kv_i4(kv, k.view(B, heads, 64), v.view(B, heads, 64), ks.view(B, heads, 2), vs.view(B, heads, 2), 1)
        changed = (pool.buf != before[0]).any(dim=-1)                # [page, layer, kv, head, slot]
        assert not changed[:, 0].any()                               # layer 0 untouched
        assert changed.sum().item() <= 2 * heads * B
        pools.append(pool)
    assert torch.equal(pools[0].buf, pools[1].buf) and torch.equal(pools[0].param, pools[1].param)


@pytest.mark.parametrize("seqlens,heads,block", [([5, 17, 32, 1, 16], 4, 16), ([1041, 1024, 1025, 48], 8, 16), ([300] * 3 + [33], 32, 32),
                                                  ([2000, 7], 4, 16), ([300, 17, 260, 33, 1, 290, 128, 64], 32, 16),
                                                  ([150] * 19 + [3], 32, 16), ([210, 97, 5, 48, 49], 5, 48), ([300] * 7 + [49], 40, 48)])
def test_decode_with_the_append_inside_equals_append_then_decode(seqlens, heads, block):
    """atom_batch_decode_append_i4 (round 6): quantising this step's FP32 k / v sums, writing them into the last token's cache slot and
    attending over the cache in ONE launch leaves the cache bytes of atom_kv_quant_append_f32 and gives the output of the two launches,
    bit for bit -- sequence lengths that end on the first / last slot of a 16-token tile and of a page, one-token sequences, KV ranges
    split over several waves (the appending wave is the last split's) and unsplit."""
    from atom_amd import ops
    from atom_amd.utils import BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4
    dev = torch.device("cuda")
    B = len(seqlens)
    g = torch.Generator(device="cuda").manual_seed(sum(seqlens))
    k32 = torch.randn((B, heads * 128), device="cuda", generator=g) * 2
    v32 = torch.randn((B, heads * 128), device="cuda", generator=g) * 3 + 0.5
    k32[0, :128] = 0.25                                               # a constant head vector: scale 0, every code 0
    q = torch.randn((B, heads, 128), device="cuda", generator=g).half()
    outs, pools = [], []
    cap = sum(-(-n // block) for n in seqlens) + 2
    for fused in (False, True):
        pool = KvPoolInt4(num_layers=2, num_heads=heads, head_dim=128, capacity=cap, block_len=block, device=dev)
        gp = torch.Generator(device="cuda").manual_seed(11)
        pool.buf.copy_(torch.randint(0, 256, pool.buf.shape, device="cuda", dtype=torch.uint8, generator=gp))
        pool.param.copy_((torch.rand(pool.param.shape, device="cuda", generator=gp) * 0.05 + 0.01).half())
        pool._free = set(range(cap))                                  # same page numbering in both runs
        kv = BatchedKvCacheInt4([KvCacheInt4(pool, n) for n in seqlens])
        if fused:
            o = ops.batch_decode_i4(q, kv, 1, append_kv=(k32, v32))
        else:
            ops.quant_append_kv_i4(kv, k32, v32, 1)
            o = ops.batch_decode_i4(q, kv, 1)
        outs.append(o)
        pools.append(pool)
    assert torch.equal(pools[0].buf, pools[1].buf) and torch.equal(pools[0].param.view(torch.int16), pools[1].param.view(torch.int16))
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


@pytest.mark.parametrize("seqlens,heads,block", [([37, 5, 16], 4, 16), ([100, 33, 64, 2, 17], 8, 32)])
def test_append_and_decode_vs_reference_cpu_implementation(seqlens, heads, block):
    """The HIP ops directly against the reference's CPU code: the cache after atom_kv_append_i4 equals the cache after
    cpu_reference::append_paged_kv_cache byte for byte; atom_batch_decode_i4 is within 2e-3 of cpu_reference::single_quantize_mha
    (FP32, std::exp / std::cos) on the gathered keys and values of every sequence."""
    from tests import ref_golden as R
    if not R.available():
        pytest.skip("oracle/_ref/libatom_ref.so not built (needs /root/reference)")
    from atom_amd import ops
    pool, cs, kv, g = _setup(seqlens, heads=heads, block=block, seed=7)
    T = sum(seqlens)
    k = torch.randint(0, 256, (T, heads, 64), device="cuda", generator=g, dtype=torch.uint8)
    v = torch.randint(0, 256, (T, heads, 64), device="cuda", generator=g, dtype=torch.uint8)
    kp = (torch.rand((T, heads, 2), device="cuda", generator=g) * 0.2 + 0.01).half()
    vp = (torch.rand((T, heads, 2), device="cuda", generator=g) * 0.2 + 0.01).half()
    ind = torch.tensor(np.cumsum([0] + seqlens), dtype=torch.int32, device="cuda")
    data, param = t2n(pool.buf).copy(), t2n(pool.param).copy()
    layer = 1
    ops.init_kv_i4(kv, k, v, kp, vp, ind, layer)
    indptr, indices, last = _np_tables(kv)
    R.append_paged_kv_i4(data, param, indptr, indices, last, t2n(k), t2n(v), t2n(kp), t2n(vp), layer, t2n(ind))
    assert np.array_equal(t2n(pool.buf), data) and np.array_equal(t2n(pool.param).view(np.uint16), param.view(np.uint16))
    q = torch.randn((len(seqlens), heads, 128), device="cuda", generator=g).half()
    o = t2n(ops.batch_decode_i4(q, kv, layer)).astype(np.float64)
    for b, S in enumerate(seqlens):
        pages = indices[indptr[b]:indptr[b + 1]]
        gather = lambda a, which: np.concatenate([a[pg, layer, which].transpose(1, 0, 2) for pg in pages], axis=0)[:S]
        want = R.single_decode_i4(t2n(q)[b], gather(data, 0), gather(data, 1), gather(param, 0), gather(param, 1))
        err = np.abs(o[b] - want).max()
        assert err <= 2e-3 * np.abs(want).max() + 1e-3, (b, err)


@pytest.mark.parametrize("max_pages", [0, 64])
def test_batch_decode_empty_sequence_gives_zeros(max_pages):
    """A sequence with no tokens (indptr[b+1] == indptr[b]) attends to nothing: its output rows are zeros -- not NaN from 0 / 0 and
    no page-table read past its (empty) page list -- and the other sequences of the batch are unaffected; both the single-pass
    kernel and the KV-split + merge path."""
    import types
    from atom_amd import ops
    pool, cs, kv, g = _setup([20, 33], heads=4, block=16, seed=11)
    q = torch.randn((3, 4, 128), device="cuda", generator=g).half()
    want = ops.batch_decode_i4(q[[0, 2]].contiguous(), kv, 1)
    indptr = t2n(kv.indptr)
    ip3 = torch.tensor([indptr[0], indptr[1], indptr[1], indptr[2]], dtype=torch.int32, device="cuda")   # sequence 1 is empty
    lpo = t2n(kv.last_page_offset)
    lp3 = torch.tensor([lpo[0], 0, lpo[1]], dtype=torch.int32, device="cuda")
    kv3 = types.SimpleNamespace(data=kv.data, param=kv.param, indptr=ip3, indicies=kv.indicies, last_page_offset=lp3, max_pages=max_pages)
    o = ops.batch_decode_i4(q, kv3, 1)
    assert torch.isfinite(o).all() and not o[1].any()
    assert torch.allclose(o[[0, 2]].float(), want.float(), atol=2e-3 * float(want.float().abs().max()), rtol=0)


def test_graph_capture_grows_the_workspace_safely():
    """The split workspace of atom_batch_decode_i4 is per (device, stream).  Two HIP graphs captured one after the other with torch's
    default capture stream share that key; the second needs a bigger workspace.  Growing it during the capture must neither fail nor
    free the buffer the first graph's launches point at: both graphs replay to the eager results afterwards."""
    from atom_amd import ops
    graphs = []
    for seqlens in ([600], [900] * 6):                      # 1 sequence, then 6: more (batch x head x split) partials
        pool, cs, kv, g = _setup(seqlens, heads=32, block=16, seed=3 + len(seqlens))
        q = torch.randn((len(seqlens), 32, 128), device="cuda", generator=g).half()
        want = ops.batch_decode_i4(q, kv, 0).clone()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = ops.batch_decode_i4(q, kv, 0)
        graphs.append((gr, out, want, pool, kv, q))
    for gr, out, want, *_ in graphs + graphs[::-1]:
        out.zero_()
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)
