"""The repo's restatement of the CUDA kernels' quantisation arithmetic ("kernel mode", oracle/atom_oracle.py) against the
reference's OWN CPU golden functions -- run_cpu_reorder_fp16_i4 (test_Reorder.cu:41-112), run_cpu_activate_fp16_i4
(test_activate.cu:41-112), run_cpu_rmsnorm_fp16_i4 (test_RMSNorm.cu:122-195) -- compiled from /root/reference into
oracle/_ref/libatom_ref.so (oracle/Makefile).  The reference's unit tests accept |int diff| <= 1 and |scale diff| <= 1e-3 on
rand() data (test_Reorder.cu:269-318); here the bound is tighter: scales within one fp16 ulp (bit-equal where no
transcendental is involved), codes +-1 on a small, stated fraction (x * (1/s) vs x / s at rounding ties)."""
import numpy as np
import pytest

from oracle import atom_oracle as O
from tests import ref_golden as R
from tests.helpers import bits16, rand_act

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref/libatom_ref.so not built (needs /root/reference)")

SHAPES = [(1, 256), (37, 4096), (16, 11008), (8, 5120)]


def _cmp(ref, got, max_flip, scale_ulps):
    d4 = np.abs(ref["q4"].astype(np.int32) - got["q4"])
    d8 = np.abs(ref["q8"].astype(np.int32) - got["q8"])
    assert d4.max() <= 1 and d8.max() <= 1
    assert (d4 > 0).mean() <= max_flip and (d8 > 0).mean() <= max_flip * 20, ((d4 > 0).mean(), (d8 > 0).mean())
    assert np.abs(bits16(ref["s4"]).astype(np.int32) - bits16(got["s4"]).astype(np.int32)).max() <= scale_ulps
    assert np.abs(bits16(ref["s8"]).astype(np.int32) - bits16(got["s8"]).astype(np.int32)).max() <= scale_ulps


def test_layout_helpers_match_reference():
    L = R.lib()
    for r in list(range(70)) + [1000, 4095, 65535]:
        assert L.ref_scale_index(r) == O.scale_index(r)
    for m in list(range(1, 70)) + [1000, 4096]:
        assert R.scale_size(m) == O.scale_size(m)


@pytest.mark.parametrize("M,H", SHAPES)
def test_reorder_kernel_mode_vs_reference_cpu_golden(M, H):
    x = rand_act(M, H, seed=M + H)
    idx = np.random.default_rng(H).permutation(H).astype(np.int16)
    ref = R.reorder(x, idx)
    got = O.reorder_quant(x, idx, "kernel", 1.0)
    _cmp(ref, got, max_flip=5e-4, scale_ulps=0)                       # scales: amax / 7 rounded to half, identical
    # packing position: low nibble = even element (PackInt4.low), and the replicated scale layout, bit for bit
    assert np.array_equal(O.pack_int4(ref["q4"]), ref["packed4"])
    assert np.array_equal(bits16(O.scales_to_ref_layout(np.ascontiguousarray(ref["s4"].T))).ravel(), bits16(ref["raw_s4"]))


@pytest.mark.parametrize("M,H", SHAPES)
def test_activate_kernel_mode_vs_reference_cpu_golden(M, H):
    g = np.random.default_rng(M * 7 + H)
    a = (g.standard_normal((M, H)) * 2).astype(np.float16)
    b = g.standard_normal((M, H)).astype(np.float16)
    _cmp(R.activate(a, b), O.silu_mul_quant(a, b, "kernel", 1.0), max_flip=2e-3, scale_ulps=1)   # expf: libm vs numpy


@pytest.mark.parametrize("M,H", SHAPES)
def test_rmsnorm_kernel_mode_vs_reference_cpu_golden(M, H):
    x = rand_act(M, H, seed=3 * M + H)
    g = np.random.default_rng(H + 1)
    w = (1 + 0.1 * g.standard_normal(H)).astype(np.float16)
    idx = g.permutation(H).astype(np.int16)
    # the golden keeps the normalised row in FP32 (test_RMSNorm.cu:140-143), the CUDA kernel and the restatement round it
    # to half first (RMSNorm.cuh:112-151): codes +-1 on < 0.2 %, scales within one fp16 ulp
    _cmp(R.rmsnorm(x, w, 1e-5, idx), O.rmsnorm_reorder_quant(x, w, 1e-5, idx, "kernel", 1.0), max_flip=2e-3, scale_ulps=1)


# ---- INT4 paged KV cache: the oracle's restatements against the reference's own CPU implementations
def _paged_state(rng, B, N, P, L, lens, extra_pages=3):
    npages = [-(-s // P) for s in lens]
    total = sum(npages) + extra_pages
    perm = rng.permutation(total)
    indptr = np.concatenate([[0], np.cumsum(npages)]).astype(np.int32)
    indices = perm[: indptr[-1]].astype(np.int32)
    last = np.array([s - (n - 1) * P for s, n in zip(lens, npages)], np.int32)
    data = rng.integers(0, 256, size=(total, L, 2, N, P, 64), dtype=np.uint8)
    param = (rng.uniform(0.01, 0.2, size=(total, L, 2, N, P, 2))).astype(np.float16)
    return data, param, indptr, indices, last


def test_kv_append_matches_reference_cpu_implementation():
    """oracle.kv_append_i4 (the restatement the HIP append kernel is checked against) == cpu_reference::append_paged_kv_cache
    (kernels/src/flashinfer/cpu_reference.h:115-169), byte for byte, prefill-style (several tokens per sequence) and decode-style."""
    rng = np.random.default_rng(3)
    B, N, P, L, layer = 4, 8, 16, 3, 1
    lens = [37, 16, 1, 50]
    for appends in ([5, 16, 1, 33], [1, 1, 1, 1]):
        data, param, indptr, indices, last = _paged_state(rng, B, N, P, L, lens)
        ai = np.concatenate([[0], np.cumsum(appends)]).astype(np.int32)
        T = int(ai[-1])
        k, v = rng.integers(0, 256, size=(T, N, 64), dtype=np.uint8), rng.integers(0, 256, size=(T, N, 64), dtype=np.uint8)
        kp, vp = rng.uniform(0.01, 0.2, size=(T, N, 2)).astype(np.float16), rng.uniform(0.01, 0.2, size=(T, N, 2)).astype(np.float16)
        d1, p1, d2, p2 = data.copy(), param.copy(), data.copy(), param.copy()
        O.kv_append_i4(d1, p1, indptr, indices, last, k, v, kp, vp, layer, append_indptr=ai)
        R.append_paged_kv_i4(d2, p2, indptr, indices, last, k, v, kp, vp, layer, ai)
        assert np.array_equal(d1, d2) and np.array_equal(p1.view(np.uint16), p2.view(np.uint16))
        assert not np.array_equal(d1, data)


def test_batch_decode_matches_reference_cpu_implementation():
    """oracle.batch_decode_i4 (FP64) vs cpu_reference::single_quantize_mha (FP32; :171-234 -> single_mha :30-108 with llama RoPE,
    :11-28), per sequence on the keys / values gathered from the pages: the two agree to FP32 accuracy."""
    rng = np.random.default_rng(4)
    B, N, P, L, layer = 3, 4, 16, 2, 1
    lens = [45, 16, 3]
    data, param, indptr, indices, last = _paged_state(rng, B, N, P, L, lens)
    q = rng.standard_normal((B, N, 128)).astype(np.float16)
    got = O.batch_decode_i4(q, data, param, indptr, indices, last, layer)
    for b in range(B):
        S = lens[b]
        pages = indices[indptr[b]:indptr[b + 1]]
        gather = lambda a, kv: np.concatenate([a[pg, layer, kv].transpose(1, 0, 2) for pg in pages], axis=0)[:S]   # [S, N, .]
        want = R.single_decode_i4(q[b], gather(data, 0), gather(data, 1), gather(param, 0), gather(param, 1))
        scale = np.abs(want).max()
        assert np.abs(got[b] - want).max() <= 2e-5 * scale + 1e-6, (b, np.abs(got[b] - want).max(), scale)


# ---- the u4-output GEMM epilogue: the restatement of the reference CODE against the reference's own local_max_min
def o4_sums(M, N, seed):
    """FP32 sums with every case the epilogue distinguishes: all-positive rows, mixed signs, a wide dynamic range, groups whose
    magnitudes nearly coincide (large quotients) and an all-equal group (scale 0)."""
    g = np.random.default_rng(seed)
    c = (g.standard_normal((M, N)) * g.uniform(0.05, 40.0, size=(M, 1))).astype(np.float32)
    c[: M // 4] = np.abs(c[: M // 4])
    if M >= 8:
        c[M // 4] = np.float32(3.0) + (g.standard_normal(N) * 1e-3).astype(np.float32)     # |x| nearly equal: quotients up to ~ 1e4
        c[M // 4 + 1, :128] = np.float32(-2.5)                                              # scale 0
    return c


@pytest.mark.parametrize("M,N", [(1, 128), (40, 512), (257, 4096)])
def test_o4_reference_code_restatement_vs_reference_local_max_min(M, N):
    """oracle.quant_o4(ref_extrema=True) -- what ATOM_O4_REF_EXTREMA is tested against on the GPU -- equals the reference's epilogue
    (DenseLayerGEMM_i4_o4.cu:722-786 around its own local_max_min, compiled from the reference source) bit for bit: (scale, zero)
    everywhere, codes wherever the reference's unclamped int8 cast is defined."""
    c = o4_sums(M, N, seed=M + N)
    want_q, want_sz = R.o4_epilogue(c)
    got_q, got_sz = O.quant_o4(c, ref_extrema=True)
    assert np.array_equal(bits16(got_sz), bits16(want_sz))
    ok = R.o4_in_range(c)
    assert ok.mean() > 0.9
    assert np.array_equal(got_q.reshape(M, -1, 64)[ok], want_q.reshape(M, -1, 64)[ok])
    # and the intended epilogue (the library's default) is the same thing on groups without negative values
    pos = (c.reshape(M, -1, 128) >= 0).all(-1) & ok
    if pos.any():
        dq, dsz = O.quant_o4(c)
        assert np.array_equal(dq.reshape(M, -1, 64)[pos], want_q.reshape(M, -1, 64)[pos])
        assert np.array_equal(bits16(dsz)[pos], bits16(want_sz)[pos])
