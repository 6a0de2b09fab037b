"""GPU parity tests of the W4A4 GEMM through the C ABI (atom_amd.ops.dense_layer_gemm_i4_fp16).

Checkers: the numpy oracle (oracle/atom_oracle.py) at sizes it finishes in seconds, the reference-generated golden
(config 1) and, at BASELINE.json's full sizes, an independent FP64 torch evaluation on the GPU plus exact algebraic
properties (scale linearity, M-tail independence, layout equivalence)."""
import os

import numpy as np
import pytest
import torch

from oracle import atom_oracle as O
from tests.helpers import (assert_gemm_close, gemm_ref_torch_f64, rand_gemm_operands, t2n, to_device, bits16)

pytestmark = pytest.mark.gpu


def _ops():
    from atom_amd import ops
    return ops


def _exact(d):
    return O.gemm_w4a4_exact(d["qa4"], d["qb4"], d["sA"], d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"])


# (M, N, K): K includes the 128 keeper columns.  Ragged M (tails), minimum K (one int4 group), N not a multiple
# of the 256-wide block tile, M = 1 (config 2 shape is covered at full size below).
SMALL = [(16, 512, 512), (1, 256, 256), (7, 256, 640), (129, 320, 384), (257, 1024, 1152), (128, 256, 4096),
         (300, 64, 1280), (5, 4096, 4096)]


@pytest.mark.parametrize("M,N,K", SMALL)
@pytest.mark.parametrize("layout", ["ref", "plain"])
def test_gemm_vs_oracle(M, N, K, layout):
    ops = _ops()
    d = rand_gemm_operands(M, N, K, seed=M + N + K)
    out = ops.dense_layer_gemm_i4_fp16(*to_device(d, layout), scale_layout=layout)
    assert out.shape == (M, N) and out.dtype == torch.float16
    assert_gemm_close(t2n(out), _exact(d), f"{M}x{N}x{K} {layout}")
    # and within the north-star 1e-2 of the reference kernel's own rounding order (fp16 scale product)
    ref = O.gemm_w4a4_ref(d["qa4"], d["qb4"], d["sA"], d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"]).astype(np.float64)
    rms = np.sqrt((ref ** 2).mean())
    assert np.abs(t2n(out).astype(np.float64) - ref).max() <= 1e-2 * rms


def test_gemm_extreme_codes():
    """All-(-8) x all-(-8) int4 and all-(-128) int8: the largest integer dot products (2^13 / 2^21); checks the
    magic-number accumulator trick at its range limit, and +7/-8 sign handling of the nibble widening."""
    ops = _ops()
    M, N, K = 64, 128, 512
    d = rand_gemm_operands(M, N, K, seed=1)
    d["qa4"][:] = -8; d["qb4"][:] = -8; d["qa8"][:] = -128; d["qb8"][:] = -128
    d["qa4"][1::2] = 7; d["qa8"][1::2] = 127
    out = ops.dense_layer_gemm_i4_fp16(*to_device(d, "plain"), scale_layout="plain")
    assert_gemm_close(t2n(out), _exact(d), "extreme codes")
    # asymmetric pattern: catches transposed / permuted operands
    d2 = rand_gemm_operands(M, N, K, seed=2)
    d2["qa4"][:] = 0; d2["qa8"][:] = 0
    d2["qa4"][3, 5] = 7        # single non-zero activation
    out2 = t2n(ops.dense_layer_gemm_i4_fp16(*to_device(d2, "plain"), scale_layout="plain")).astype(np.float64)
    ex = _exact(d2)
    assert_gemm_close(out2, ex, "single nonzero")
    assert np.count_nonzero(out2[np.arange(M) != 3]) == 0


def test_config1_golden_end_to_end(golden_dir):
    """BASELINE config 1 (M=16 N=512 K=512): HIP weight packer + HIP act quant (sim) + HIP GEMM against the
    reference's own QLinearLayer.forward output, tolerance 1e-2 relative (north star)."""
    ops = _ops()
    z = np.load(os.path.join(golden_dir, "c1_qlinear_16x512x512.npz"))
    b4, b8, sb, sb8 = ops.quant_weight_w4(torch.from_numpy(z["W"]).cuda(), 0.85, 2)
    o8, o4, s8, s4, xq = ops.reorder_fp16_i4(torch.from_numpy(z["x"]).cuda(), None, quant_mode="sim", clip=0.9,
                                             scale_layout="plain", return_dequant=True)
    assert np.array_equal(bits16(t2n(xq)), bits16(z["xq"]))             # fake-quant activation bit-exact
    y = t2n(ops.dense_layer_gemm_i4_fp16(o4, b4, s4, sb, o8, b8, s8, sb8, scale_layout="plain")).astype(np.float64)
    want = z["y"].astype(np.float64)
    rms = np.sqrt((want ** 2).mean())
    assert np.abs(y - want).max() <= 1e-2 * rms
    assert np.linalg.norm(y - want) / np.linalg.norm(want) < 1e-3


C5 = [(m, n, k) for (n, k) in ((5120, 5120), (13824, 5120), (5120, 13824)) for m in (1, 16, 64, 256, 2048)]   # config 5: all 15
FULL = [(1, 4096, 4096),            # config 2
        (4096, 4096, 4096),         # config 3 (headline)
        *C5,
        (2048, 11008, 4096), (1000, 4096, 11008)]                                                         # config 4 shapes


@pytest.mark.parametrize("M,N,K", FULL)
def test_gemm_full_size_vs_fp64(M, N, K):
    ops = _ops()
    d = rand_gemm_operands(M, N, K, seed=M ^ N ^ K)
    dev = to_device(d, "plain")
    out = ops.dense_layer_gemm_i4_fp16(*dev, scale_layout="plain")
    ref = gemm_ref_torch_f64(d)
    assert_gemm_close(t2n(out), t2n(ref), f"full {M}x{N}x{K}")


def test_gemm_properties_full_size():
    """Size-independent exact properties at the headline shape (no oracle needed):
       * doubling every activation scale doubles D exactly (powers of two commute with every rounding);
       * rows are independent: computing only the first tokens reproduces those rows bit-for-bit (among batch sizes that get
         the same summation order, atom_gemm_w4a4_f6_order);
       * the replicated and the plain A-scale layouts give identical bits."""
    ops = _ops()
    M, N, K = 4096, 4096, 4096
    d = rand_gemm_operands(M, N, K, seed=11)
    a = to_device(d, "plain")
    base = ops.dense_layer_gemm_i4_fp16(*a, scale_layout="plain")
    a2 = list(a); a2[2] = a[2] * 2; a2[6] = a[6] * 2
    dbl = ops.dense_layer_gemm_i4_fp16(*a2, scale_layout="plain")
    normal = base.abs() >= 2.0 ** -13           # fp16 subnormal results carry fewer bits: 2*round(c) != round(2c)
    assert torch.equal(dbl[normal], (base * 2)[normal])
    assert (dbl.float() - 2 * base.float()).abs().max() <= 2.0 ** -23
    rows = lambda Mh: ops.dense_layer_gemm_i4_fp16(a[0][:Mh].contiguous(), a[1], a[2][:, :Mh].contiguous(), a[3],
                                                   a[4][:Mh].contiguous(), a[5], a[6][:Mh].contiguous(), a[7], scale_layout="plain")
    order = ops.L.lib().atom_gemm_w4a4_f6_order
    assert order(M, N, K) == order(2500, N, K) == 1 and order(1000, N, K) == order(900, N, K) == 2
    assert torch.equal(rows(2500), base[:2500])                 # same summation order (K steps in order): same bits
    sub = rows(1000)                                            # mid-size batches (packed operands are re-coded to F6 from 257
    assert torch.equal(rows(900), sub[:900])                    # rows): two ordered halves of the K steps
    assert_gemm_close(t2n(sub), t2n(base[:1000]).astype(np.float64), "rows 0..999 in the other summation order")
    r = to_device(d, "ref")
    assert torch.equal(ops.dense_layer_gemm_i4_fp16(*r, scale_layout="ref"), base)


def test_gemm_error_behaviour():
    ops = _ops()
    from atom_amd._lib import AtomHipError
    d = rand_gemm_operands(4, 64, 256, seed=0)
    a = to_device(d, "plain")
    ops.dense_layer_gemm_i4_fp16(*a, scale_layout="plain")                       # smallest legal problem
    bad = list(a); bad[1] = a[1][:60].contiguous()                                # N = 60: not a multiple of 64
    with pytest.raises(AtomHipError):
        ops.dense_layer_gemm_i4_fp16(*bad, scale_layout="plain")
    cpu = [t.cpu() for t in a]
    with pytest.raises(AtomHipError):
        ops.dense_layer_gemm_i4_fp16(*cpu, scale_layout="plain")


def test_gemm_bit_exact_vs_c_contract():
    """The arithmetic contract of include/atom_hip.h restated in plain C (oracle/atom_oracle.c): exact integer dots,
    c = fmaf((float)idot, sA * sB, c) with the scale product exact in FP32, groups in order, then the keeper (one dot product over its 128 columns, one
    de-quantisation), D = half(c).  Bit-for-bit -- in the summation order of the kernel the shape is dispatched to: one ordered sum
    for the tile kernels, the G + 1 items dealt to 8 waves for decode batches (2 <= M <= 256 where the decode-batch kernel fits)."""
    from tests import c_oracle as C
    ops = _ops()
    for (M, N, K) in [(40, 128, 640), (257, 64, 384), (3, 320, 1152), (300, 128, 640), (600, 192, 640)]:      # (short K: no split-K through the workspace)
        d = rand_gemm_operands(M, N, K, seed=M * 7 + N)
        out = t2n(ops.dense_layer_gemm_i4_fp16(*to_device(d, "plain"), scale_layout="plain"))
        order = ops.L.lib().atom_gemm_w4a4_packed_order(M, N, K, 1)                 # the library's own statement of its dispatch
        assert order in (1, 8)
        want = C.gemm(O.pack_int4(d["qa4"]), O.pack_int4(d["qb4"]), d["sA"].T, d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"], nsplit=order)
        assert np.array_equal(bits16(out), bits16(want)), f"{M}x{N}x{K}: {(bits16(out) != bits16(want)).sum()} elements differ"


@pytest.mark.parametrize("M,N,K", [(40, 128, 640), (300, 384, 384), (64, 4096, 1152)])
def test_gemm_o4_epilogue(M, N, K):
    """dense_layer_gemm_i4_o4: u4 codes + (scale, zero) per 128-column output group, bit-exact against the oracle's
    restatement of the epilogue applied to the C-contract FP32 accumulators; and the de-quantised tensor
    q*scale - zero (flashinfer/quantization.cuh:59-84) is within half a quantisation step of the fp16 GEMM."""
    from tests import c_oracle as C
    ops = _ops()
    d = rand_gemm_operands(M, N, K, seed=M + 3 * N)
    dev = to_device(d, "plain")
    q, sz = ops.dense_layer_gemm_i4_o4(*dev, scale_layout="plain", use_workspace=False)      # the tile kernel
    assert q.shape == (M, N // 2) and q.dtype == torch.uint8 and sz.shape == (M, N // 128 * 2)
    c32 = C.gemm(O.pack_int4(d["qa4"]), O.pack_int4(d["qb4"]), d["sA"].T, d["sB"], d["qa8"], d["qb8"], d["sA8"],
                 d["sB8"], fp32=True)
    want_q, want_sz = O.quant_o4(c32)
    assert np.array_equal(bits16(t2n(sz).reshape(M, N // 128, 2)), bits16(want_sz))
    assert np.array_equal(t2n(q), want_q)
    # round trip
    codes = np.stack([t2n(q) & 0xF, t2n(q) >> 4], axis=-1).reshape(M, N).astype(np.float32)
    s = t2n(sz).reshape(M, N // 128, 2).astype(np.float32)
    deq = codes.reshape(M, N // 128, 128) * s[..., 0:1] - s[..., 1:2]
    ref = t2n(ops.dense_layer_gemm_i4_fp16(*dev, scale_layout="plain")).astype(np.float32).reshape(M, N // 128, 128)
    assert np.abs(deq - ref).max() <= 0.51 * s[..., 0].max() + 0.1


@pytest.mark.parametrize("M,N,K,ws", [(40, 128, 640, False), (300, 384, 384, False), (40, 128, 640, True), (16, 4096, 1152, True)])
def test_gemm_o4_reference_extrema_mode(M, N, K, ws):
    """ATOM_O4_REF_EXTREMA: the u4 epilogue as the reference CODE computes it (extrema of |x|, 4-bit wrap;
    DenseLayerGEMM_i4_o4.cu:73-80, :749-771), bit-exact against the oracle's restatement of that code -- on the tile kernel
    and on the decode path -- and how far it is from the intended min / max epilogue (the default): identical on groups
    without negative values, different codes on most elements otherwise, which is why the default does not follow it."""
    from tests import c_oracle as C
    ops = _ops()
    d = rand_gemm_operands(M, N, K, seed=M + 5 * N + K)
    d["qa4"][: M // 2] = np.abs(d["qa4"][: M // 2]); d["qa8"][: M // 2] = np.abs(d["qa8"][: M // 2].astype(np.int16)).clip(0, 127).astype(np.int8)
    d["qb4"] = np.abs(d["qb4"]); d["qb8"] = np.abs(d["qb8"].astype(np.int16)).clip(0, 127).astype(np.int8)   # first half of the rows: sums >= 0
    dev = to_device(d, "plain")
    q_ref, sz_ref = ops.dense_layer_gemm_i4_o4(*dev, scale_layout="plain", use_workspace=ws, ref_extrema=True)
    q_def, sz_def = ops.dense_layer_gemm_i4_o4(*dev, scale_layout="plain", use_workspace=ws)
    if ws:
        assert ops.decode_gemm_fits(M, N, K)
        c32 = t2n(ops.dense_layer_gemm_i4_f32(*dev, scale_layout="plain"))
    else:
        c32 = C.gemm(O.pack_int4(d["qa4"]), O.pack_int4(d["qb4"]), d["sA"].T, d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"], fp32=True)
    want_q, want_sz = O.quant_o4(c32, ref_extrema=True)
    assert np.array_equal(bits16(t2n(sz_ref).reshape(M, N // 128, 2)), bits16(want_sz))
    assert np.array_equal(t2n(q_ref), want_q)
    h = M // 2
    assert (c32[:h] >= 0).all() and (c32[h:] < 0).any()
    assert torch.equal(q_ref[:h], q_def[:h]) and torch.equal(sz_ref[:h], sz_def[:h])       # no negative values: same epilogue
    nib = lambda t: np.stack([t2n(t) & 0xF, t2n(t) >> 4], axis=-1).reshape(t.shape[0], -1)
    differ = (nib(q_ref[h:]) != nib(q_def[h:])).mean()
    assert differ > 0.5, differ                                                               # mixed signs: most codes differ


@pytest.mark.parametrize("M,N,K", [(8, 4096, 4096), (17, 1408, 2176), (64, 5120, 5120), (100, 256, 4096), (256, 5120, 1152)])
def test_gemm_split_k_path(M, N, K):
    """Skinny shapes go through atom_gemm_w4a4_f16_ws (split-K over FP32 partials + reduce launch) when the policy
    grants a workspace; same tolerance as the unsplit kernels, and the plain entry point agrees to fp16 rounding."""
    from atom_amd import _lib as L
    ops = _ops()
    lib = L.lib()
    d = rand_gemm_operands(M, N, K, seed=M * 3 + K)
    dev = to_device(d, "plain")
    out = ops.dense_layer_gemm_i4_fp16(*dev, scale_layout="plain")            # uses the workspace variant
    assert_gemm_close(t2n(out), gemm_ref_torch_f64(d).cpu().numpy(), f"split-K {M}x{N}x{K}")
    plain = torch.empty_like(out)
    st = lib.atom_gemm_w4a4_f16(*[t.data_ptr() for t in dev], plain.data_ptr(), M, N, K, 128, 128, 1,
                                torch.cuda.current_stream().cuda_stream)
    assert st == 0
    a, b = t2n(out).astype(np.float64), t2n(plain).astype(np.float64)
    assert (np.abs(a - b) <= 1e-3 * np.abs(b) + 1e-3 * np.sqrt((b ** 2).mean())).all()    # different FP32 summation order
    if lib.atom_gemm_w4a4_workspace_bytes(M, N, K) == 0:
        assert np.array_equal(bits16(t2n(out)), bits16(t2n(plain)))


@pytest.mark.parametrize("M,N,K", [(1, 128, 256), (16, 4096, 4096), (40, 128, 640), (64, 4096, 1152), (33, 512, 11008),
                                   (128, 4096, 4096), (200, 256, 1152)])
@pytest.mark.parametrize("layout", ["ref", "plain"])
def test_gemm_o4_decode_path(M, N, K, layout):
    """atom_gemm_w4a4_o4_ws: decode batches run the weight-streaming kernel (FP32 sums into the workspace) + a u4
    epilogue launch.  Same epilogue arithmetic as the tile kernel on sums that differ in the last FP32 bits (wave-order
    summation): (scale, zero) within one fp16 ulp, codes equal except where a value sits on a rounding boundary (then
    off by one), de-quantised tensor within half a step of the fp16 GEMM."""
    ops = _ops()
    assert ops.L.lib().atom_gemm_w4a4_o4_workspace_bytes(M, N, K) == 4 * M * N
    d = rand_gemm_operands(M, N, K, seed=M + 7 * N)
    dev = to_device(d, layout)
    q, sz = ops.dense_layer_gemm_i4_o4(*dev, scale_layout=layout)
    q_t, sz_t = ops.dense_layer_gemm_i4_o4(*dev, scale_layout=layout, use_workspace=False)
    a, b = t2n(sz).astype(np.float64), t2n(sz_t).astype(np.float64)
    assert (np.abs(a - b) <= 2e-3 * np.abs(b) + 1e-6).all()
    ca = np.stack([t2n(q) & 0xF, t2n(q) >> 4], axis=-1).reshape(M, N).astype(np.int32)
    cb = np.stack([t2n(q_t) & 0xF, t2n(q_t) >> 4], axis=-1).reshape(M, N).astype(np.int32)
    assert np.abs(ca - cb).max() <= 1 and (ca != cb).mean() < 0.02
    s = t2n(sz).reshape(M, N // 128, 2).astype(np.float32)
    deq = ca.astype(np.float32).reshape(M, N // 128, 128) * s[..., 0:1] - s[..., 1:2]
    ref = t2n(ops.dense_layer_gemm_i4_fp16(*dev, scale_layout=layout)).astype(np.float32).reshape(M, N // 128, 128)
    assert np.abs(deq - ref).max() <= 0.51 * s[..., 0].max() + 0.1
    assert torch.equal(q, ops.dense_layer_gemm_i4_o4(*dev, scale_layout=layout)[0])              # deterministic


# decode batches: gemm_w4a4_skinny.hip (2 <= M <= 256; 8 waves split K: 1..14 groups per wave incl. waves with no work,
# keeper on the last working wave, 1 / 2 / 4 / 8 / 16 token blocks, ragged M, N = 64 .. 13824)
SKINNY = [(2, 256, 256), (7, 4096, 4096), (8, 64, 384), (16, 512, 512), (17, 1408, 2176), (31, 320, 1152), (33, 5120, 5120),
          (48, 11008, 4096), (64, 4096, 11008), (16, 5120, 13824), (64, 1024, 1280), (65, 4096, 4096), (100, 1408, 2176),
          (128, 5120, 5120), (129, 320, 384), (200, 1408, 2176), (256, 4096, 4096)]


@pytest.mark.parametrize("M,N,K", SKINNY)
@pytest.mark.parametrize("layout", ["ref", "plain"])
def test_gemm_decode_batches(M, N, K, layout):
    ops = _ops()
    d = rand_gemm_operands(M, N, K, seed=M * 13 + N + K)
    t = to_device(d, layout)
    out = ops.dense_layer_gemm_i4_fp16(*t, scale_layout=layout)
    exact = gemm_ref_torch_f64(d).cpu().numpy() if N * K > (1 << 24) else _exact(d)
    assert_gemm_close(t2n(out), exact, f"decode batch {M}x{N}x{K} {layout}")
    assert torch.equal(out, ops.dense_layer_gemm_i4_fp16(*t, scale_layout=layout))          # deterministic summation order
    if M <= 16:                                              # rows are independent: a batch equals its rows one by one
        one = to_device({k: (v[M - 1:M] if k in ("qa4", "qa8", "sA", "sA8") else v) for k, v in d.items()}, layout)
        row = ops.dense_layer_gemm_i4_fp16(*one, scale_layout=layout)
        assert_gemm_close(t2n(row), exact[M - 1:M], "last row alone")


@pytest.mark.parametrize("M,N,K", [(2, 256, 256), (8, 64, 384), (16, 512, 512), (17, 1408, 2176), (31, 320, 1152),
                                   (64, 1024, 1280), (100, 1408, 2176), (129, 320, 384), (16, 256, 13824)])
def test_gemm_decode_batches_bit_exact_vs_c_contract(M, N, K):
    """The decode-batch kernel is bit-identical to the C restatement of ITS summation order (oracle/atom_oracle.c,
    nsplit = 8: the G + 1 items dealt to 8 waves in consecutive slices, per-item arithmetic of the contract, partial sums
    added in wave order)."""
    from tests import c_oracle as C
    ops = _ops()
    d = rand_gemm_operands(M, N, K, seed=M * 17 + N + K)
    out = ops.dense_layer_gemm_i4_fp16(*to_device(d, "plain"), scale_layout="plain")
    want = C.gemm(O.pack_int4(d["qa4"]), O.pack_int4(d["qb4"]), d["sA"].T, d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"],
                  nsplit=8)
    assert np.array_equal(bits16(t2n(out)), bits16(want))


def test_gemm_decode_batches_random_shapes_bit_exact():
    """24 seeded random decode shapes (M 2..256, N 64..1024, K 256..3712, both scale layouts) against the C restatement
    of the kernel's summation order: every slice length 1..4 and 5..8 per wave, idle waves, ragged token blocks."""
    from tests import c_oracle as C
    ops = _ops()
    g = np.random.default_rng(2024)
    for i in range(24):
        M = int(g.integers(2, 257))
        N = 64 * int(g.integers(1, 17))
        K = 128 * int(g.integers(2, 30))
        if ops.L.lib().atom_gemm_w4a4_o4_workspace_bytes(M, (N + 127) // 128 * 128, K) == 0:
            continue                                         # not a decode-kernel shape (the policy keeps the tile kernels)
        layout = ("ref", "plain")[i & 1]
        d = rand_gemm_operands(M, N, K, seed=1000 + i)
        out = ops.dense_layer_gemm_i4_fp16(*to_device(d, layout), scale_layout=layout)
        want = C.gemm(O.pack_int4(d["qa4"]), O.pack_int4(d["qb4"]), d["sA"].T, d["sB"], d["qa8"], d["qb8"], d["sA8"],
                      d["sB8"], nsplit=8)
        assert np.array_equal(bits16(t2n(out)), bits16(want)), (M, N, K, layout)


@pytest.mark.parametrize("M,N,K", [(16, 512, 512), (129, 320, 384), (257, 1024, 1152), (300, 64, 1280), (8, 4096, 4096),
                                   (64, 5120, 5120), (1024, 1024, 2176), (5, 256, 640)])
@pytest.mark.parametrize("layout", ["ref", "plain"])
def test_gemm_wide_activations_bit_identical(M, N, K, layout):
    """ATOM_A_WIDE: the activation operand pre-widened to int8 (x16, even/odd de-interleaved) is the same arithmetic in
    the same order: results equal the packed call bit for bit wherever both run the same kernel family (MFMA tiles:
    M > 256), and match the exact oracle everywhere (ragged M / N tails, split-K shapes, M <= 7 on the tile kernel)."""
    from tests.helpers import wide_codes
    ops = _ops()
    d = rand_gemm_operands(M, N, K, seed=M * 5 + N + K)
    t = to_device(d, layout)
    aw = torch.from_numpy(wide_codes(d["qa4"])).cuda()
    out_w = ops.dense_layer_gemm_i4_fp16(aw, *t[1:], scale_layout=layout, a_wide=True)
    assert_gemm_close(t2n(out_w), _exact(d), f"wide {M}x{N}x{K} {layout}")
    if M > 256:                                         # same tile family (M <= 256: the decode kernels may run, in wave order)
        out_p = ops.dense_layer_gemm_i4_fp16(*t, scale_layout=layout)
        if ops.L.lib().atom_gemm_w4a4_workspace_bytes(M, N, K) == 0:
            assert torch.equal(out_w, out_p)


def test_gemm_wide_full_size_and_errors():
    from tests.helpers import wide_codes
    from atom_amd._lib import AtomHipError
    ops = _ops()
    d = rand_gemm_operands(4096, 4096, 4096, seed=11)
    t = to_device(d, "plain")
    aw = torch.from_numpy(wide_codes(d["qa4"])).cuda()
    out_w = ops.dense_layer_gemm_i4_fp16(aw, *t[1:], scale_layout="plain", a_wide=True)
    out_p = ops.dense_layer_gemm_i4_fp16(*t, scale_layout="plain")
    assert torch.equal(out_w, out_p)                      # both: per-group FMAs in group order, the keeper last
    with pytest.raises(AtomHipError):                     # the u4-epilogue kernel takes packed activations only
        st = ops.L.lib().atom_gemm_w4a4_o4(aw.data_ptr(), *[x.data_ptr() for x in t[1:]], t[0].data_ptr(),
                                           t[2].data_ptr(), 4096, 4096, 4096, 128, 128, 1 | ops.L.A_WIDE, None)
        ops.L.check(st, "atom_gemm_w4a4_o4")


# (1500, 1408, 640) and (2048, 2048, 1152) run the 128x128 geometry (ragged / whole tiles), the others 64x128
@pytest.mark.parametrize("M,N,K", [(16, 512, 512), (257, 1024, 1152), (300, 64, 1280), (1024, 1024, 2176), (5, 256, 640),
                                   (1500, 1408, 640), (2048, 2048, 1152)])
def test_gemm_f6_operands_bit_identical(M, N, K):
    """ATOM_AB_F6: both operands as BF6 streams on the block-scaled MFMA (gemm_w4a4_f6.hip).  Integer dot products are
    exact there too and the FP32 de-quantisation order is the same: equal to the packed call bit for bit where that one
    runs the MFMA tile kernel without split-K, and to the exact oracle everywhere.  The weight comes from
    atom_repack_weight_f6, the activation buffer is built on the host here (the quantiser path is tested below)."""
    from tests.helpers import f6_codes
    ops = _ops()
    d = rand_gemm_operands(M, N, K, seed=M * 11 + N + K)
    t = to_device(d, "plain")
    a6 = torch.from_numpy(f6_codes(d["qa4"], d["sA"])).cuda()
    b6 = ops.repack_weight_f6(t[1])
    assert np.array_equal(t2n(b6), f6_codes(d["qb4"]))
    out = ops.dense_layer_gemm_i4_fp16(a6, b6, *t[2:], scale_layout="plain", a_wide="f6")
    assert_gemm_close(t2n(out), _exact(d), f"f6 {M}x{N}x{K}")
    order = ops.L.lib().atom_gemm_w4a4_f6_order(M, N, K)
    if order > 1:                                         # two / four K groups per tile: ordered ranges of the K steps, summed
        from tests import c_oracle
        want = c_oracle.gemm(O.pack_int4(d["qa4"]), O.pack_int4(d["qb4"]), np.ascontiguousarray(d["sA"].T), d["sB"],
                             d["qa8"], d["qb8"], d["sA8"], d["sB8"], nsplit=-order)
        assert np.array_equal(bits16(t2n(out)), bits16(want))
    elif M > 256 and ops.L.lib().atom_gemm_w4a4_workspace_bytes(M, N, K) == 0:
        assert order == 1
        assert torch.equal(out, ops.dense_layer_gemm_i4_fp16(*t, scale_layout="plain"))


# mid-size prefill batches in the BF6 format: up to 512 tiles of 64x64 the mid-size-batch kernel (gemm_w4a4_mid.hip: the K steps in
# order); beyond, while at most 256 tiles of 128x128 exist, 128x128 tiles shared by two groups of 4 waves that split the K steps.  Whole
# and ragged tiles in both dimensions, odd and even step counts, with and without the appended fp32 weight scales.
@pytest.mark.parametrize("M,N,K,want_cfg", [(1024, 4096, 4096, "128x128"), (512, 4096, 4096, "mid"), (300, 1088, 1152, "mid"),
                                            (700, 2112, 2176, "mid"), (1000, 3136, 1408, "128x128"), (257, 10880, 1024, "128x128"),
                                            (2048, 2048, 1152, "128x128"), (129, 4096, 4096, "mid"), (70, 13824, 640, "mid")])
@pytest.mark.parametrize("f6s", [True, False])
def test_gemm_f6_two_k_group_kernels(M, N, K, want_cfg, f6s):
    """Bit for bit against the C restatement in the kernel's summation order -- the K steps in order (the mid-size-batch kernel) or in
    two ordered ranges (oracle gemm_core, nsplit = -2) -- on rows from the first, a middle and the last tile; the whole output against
    the exact value."""
    from tests import c_oracle
    from tests.helpers import f6_codes
    ops = _ops()
    lib = ops.L.lib()
    t64, t128 = -(-M // 64) * (N // 64), -(-M // 128) * -(-N // 128)
    assert (t64 <= 512) == (want_cfg == "mid") and (want_cfg == "mid" or t128 <= 256)   # the dispatch rule this case is meant to hit
    order = lib.atom_gemm_w4a4_f6_order(M, N, K)
    assert order == (1 if want_cfg == "mid" else 2)
    d = rand_gemm_operands(M, N, K, seed=5 * M + N + 3 * K)
    t = to_device(d, "plain")
    a6 = torch.from_numpy(f6_codes(d["qa4"], d["sA"])).cuda()
    b6 = ops.repack_weight_f6(t[1], t[3]) if f6s else ops.repack_weight_f6(t[1])
    out = ops.dense_layer_gemm_i4_fp16(a6, b6, *t[2:], scale_layout="plain", a_wide="f6")
    assert_gemm_close(t2n(out), _exact(d), f"f6 two K groups {M}x{N}x{K}")
    rows = np.unique(np.r_[0:6, M // 2 - 3:M // 2 + 3, M - 6:M])
    want = c_oracle.gemm(O.pack_int4(d["qa4"][rows]), O.pack_int4(d["qb4"]), np.ascontiguousarray(d["sA"][rows].T), d["sB"],
                         d["qa8"][rows], d["qb8"], d["sA8"][rows], d["sB8"], nsplit=-order if order > 1 else 1)
    assert np.array_equal(bits16(t2n(out)[rows]), bits16(want))
    # the plain entry point takes the same route (no workspace involved)
    D = torch.empty_like(out)
    flags = ops.L.SCALE_LAYOUT_PLAIN | ops.L.AB_F6 | (ops.L.B_F6S if f6s else 0) | (ops.L.B_SCALE_PAIRS if getattr(b6, "atom_pairs", False) else 0)
    bbuf = b6.atom_f6s if f6s else b6
    st = lib.atom_gemm_w4a4_f16(a6.data_ptr(), bbuf.data_ptr(), *[x.data_ptr() for x in t[2:]], D.data_ptr(), M, N, K, 128, 128, flags,
                                ops.L.current_stream(D.device))
    ops.L.check(st, "atom_gemm_w4a4_f16")
    assert torch.equal(D, out)


@pytest.mark.parametrize("M,K,layout", [(64, 5120, "plain"), (1, 1152, "plain"), (200, 4224, "ref"), (2048, 1152, "plain"), (2049, 1408, "ref"),
                                        (4096, 4096, "plain")])
def test_repack_kernels_write_the_f6_records_bit_for_bit(M, K, layout):
    """atom_repack_act_f6 (the 64-row-block kernel up to 2,048 rows, the 256-row LDS kernel beyond) and atom_repack_weight_f6s against
    the numpy restatement of the record format (tests/helpers.f6_codes): code fields, the fp16 scale at byte 96, the same value as
    float32 at byte 100, pad rows zero in the fields the GEMM reads; the weight's appended float32 scales."""
    from tests.helpers import f6_codes, f6_fields
    ops = _ops()
    N = 320
    d = rand_gemm_operands(M, N, K, seed=M + K)
    t = to_device(d, layout)
    a6 = t2n(ops.repack_act_f6(t[0].view(torch.uint8), t[2], scale_layout=layout))
    want = f6_codes(d["qa4"], d["sA"])
    assert a6.shape == want.shape
    assert np.array_equal(f6_fields(a6[:, :M]), f6_fields(want[:, :M])) and np.array_equal(a6[:, :M, 96:98], want[:, :M, 96:98])
    assert np.array_equal(a6[:, :M, 100:104], want[:, :M, 100:104])
    b6 = ops.repack_weight_f6(t[1], t[3])
    wb = f6_codes(d["qb4"])
    assert np.array_equal(f6_fields(t2n(b6)[:, :N]), f6_fields(wb[:, :N]))
    G, rp = wb.shape[0], wb.shape[1]
    s32 = t2n(b6.atom_f6s)[G * rp * 104:].view(np.float32).reshape(G, rp)
    assert np.array_equal(s32[:, :N], d["sB"].astype(np.float32))


def test_gemm_f6_full_size_from_the_quantiser():
    """Headline shape end to end in the native format: activation quantiser (ATOM_QUANT_F6_CODES) -> F6 GEMM equals
    activation quantiser (packed) -> INT8 GEMM, bit for bit."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(21)
    M = N = K = 4096
    x = (torch.randn((M, K), device="cuda", generator=g) * 1.5).half()
    W = (torch.randn((N, K), device="cuda", generator=g) * 0.05).half()
    b4, b8, sb, sb8 = ops.quant_weight_w4(W, 0.85, 2)
    p = ops.reorder_fp16_i4(x, None, quant_mode="sim", clip=0.9, scale_layout="plain")
    f = ops.reorder_fp16_i4(x, None, quant_mode="sim", clip=0.9, scale_layout="plain", wide_codes="f6")
    assert torch.equal(p[0], f[0]) and torch.equal(p[2], f[2]) and torch.equal(p[3], f[3])
    from tests.helpers import f6_codes, f6_fields
    want6 = f6_codes(O.unpack_int4(t2n(p[1]).view(np.uint8)), t2n(p[3]).T)
    got6 = t2n(f[1])
    # same codes (a value that rounded to zero from below is stored as BF6 -0: same number) and same in-row scales
    assert np.array_equal(f6_fields(got6[:, :M]), f6_fields(want6[:, :M]))
    assert np.array_equal(got6[:, :M, 96:], want6[:, :M, 96:])
    ref = ops.dense_layer_gemm_i4_fp16(p[1], b4, p[3], sb, p[0], b8, p[2], sb8, scale_layout="plain")
    out = ops.dense_layer_gemm_i4_fp16(f[1], ops.repack_weight_f6(b4), f[3], sb, f[0], b8, f[2], sb8,
                                       scale_layout="plain", a_wide="f6")
    assert torch.equal(out, ref)


def _gemm_plain_entry(t, M, N, K, layout_flags):
    """atom_gemm_w4a4_f16 itself (NO workspace): packed operands of prefill size run the INT8 MFMA tile kernels here, never the
    F6 kernels (atom_amd.ops routes them through the workspace entry point, which re-codes to F6)."""
    ops = _ops()
    D = torch.empty((M, N), dtype=torch.float16, device="cuda")
    st = ops.L.lib().atom_gemm_w4a4_f16(*[x.data_ptr() for x in t], D.data_ptr(), M, N, K, 128, 128, layout_flags,
                                        ops.L.current_stream(D.device))
    ops.L.check(st, "atom_gemm_w4a4_f16")
    return D


# the 256x256 F6 kernel with appended fp32 weight scales (ATOM_B_F6S, "q" kernel): whole and ragged tiles, G = 1, 2, 4, 7, 31
# int4 groups (every tail variant of its unrolled-by-3 K loop)
@pytest.mark.parametrize("M,N,K", [(4096, 4096, 4096), (4096, 4096, 256), (4096, 4096, 384), (4096, 4096, 640),
                                   (4096, 4096, 1024), (4000, 4096, 512), (4096, 4032, 768), (8192, 2048, 896),
                                   (40000, 64, 512), (36900, 192, 384)])          # tall and narrow: a quarter / three quarters of one tile column
def test_gemm_f6s_headline_kernel_vs_int8_kernel(M, N, K):
    """Two different kernel families on the same operands, bit for bit: the F6 256x256 kernel (BF6 MFMA, weight scales as the
    fp32 array atom_repack_weight_f6s appends) against the INT8 MFMA tile kernel behind the plain entry point -- and both against
    the C restatement of the arithmetic contract on a sample of rows."""
    from tests import c_oracle
    from tests.helpers import f6_codes
    ops = _ops()
    d = rand_gemm_operands(M, N, K, seed=M + 3 * N + 7 * K)
    t = to_device(d, "plain")
    a6 = torch.from_numpy(f6_codes(d["qa4"], d["sA"])).cuda()
    b6 = ops.repack_weight_f6(t[1], t[3])
    assert getattr(b6, "atom_f6s", None) is not None
    G = (K - 128) // 128
    sb32 = b6.atom_f6s[b6.numel():].view(torch.float32).view(G, -1)
    assert torch.equal(sb32[:, :N], t[3].float()) and not sb32[:, N:].any()              # appended scales: exact fp32 copies, pad zero
    out = ops.dense_layer_gemm_i4_fp16(a6, b6, *t[2:], scale_layout="plain", a_wide="f6")
    int8 = _gemm_plain_entry(t, M, N, K, ops.L.SCALE_LAYOUT_PLAIN)
    assert torch.equal(out, int8)
    rows = np.r_[0:8, M // 2:M // 2 + 8, M - 8:M]
    want = c_oracle.gemm(O.pack_int4(d["qa4"][rows]), O.pack_int4(d["qb4"]), np.ascontiguousarray(d["sA"][rows].T), d["sB"],
                         d["qa8"][rows], d["qb8"], d["sA8"][rows], d["sB8"])
    assert np.array_equal(bits16(t2n(out)[rows]), bits16(want))


def test_gemm_f6_random_shapes_bit_exact_vs_c_contract():
    """40 seeded random shapes through the F6 route (every tile geometry, ragged tiles, 1..40 int4 groups, weights with and without the
    appended fp32 scales): bit for bit against the C restatement in the summation order atom_gemm_w4a4_f6_order names, on a sample
    of rows; the whole output against the exact value."""
    from tests import c_oracle
    from tests.helpers import f6_codes
    ops = _ops()
    lib = ops.L.lib()
    rng = np.random.default_rng(2024)
    seen = set()
    for it in range(40):
        M = int(rng.choice([rng.integers(1, 64), rng.integers(64, 400), rng.integers(400, 1300), rng.integers(1300, 2600),
                            rng.integers(2600, 4200)]))
        N = 64 * int(rng.integers(1, 80))
        K = 128 * int(rng.choice([rng.integers(2, 6), rng.integers(6, 20), rng.integers(20, 42)]))
        f6s = bool(it & 1)
        order = lib.atom_gemm_w4a4_f6_order(M, N, K)
        assert order in (1, 2, 4)
        seen.add(order)
        d = rand_gemm_operands(M, N, K, seed=1000 + it)
        t = to_device(d, "plain")
        a6 = torch.from_numpy(f6_codes(d["qa4"], d["sA"])).cuda()
        b6 = ops.repack_weight_f6(t[1], t[3]) if f6s else ops.repack_weight_f6(t[1])
        out = ops.dense_layer_gemm_i4_fp16(a6, b6, *t[2:], scale_layout="plain", a_wide="f6")
        assert_gemm_close(t2n(out), _exact(d), f"f6 random {M}x{N}x{K} f6s={f6s}")
        rows = np.unique(np.clip(np.r_[0:4, M // 2:M // 2 + 4, M - 4:M], 0, M - 1))
        want = c_oracle.gemm(O.pack_int4(d["qa4"][rows]), O.pack_int4(d["qb4"]), np.ascontiguousarray(d["sA"][rows].T), d["sB"],
                             d["qa8"][rows], d["qb8"], d["sA8"][rows], d["sB8"], nsplit=-order if order > 1 else 1)
        assert np.array_equal(bits16(t2n(out)[rows]), bits16(want)), (M, N, K, f6s, order)
    assert seen == {1, 2}          # (order 4 -- four K groups on 64x128 tiles -- left the dispatch in round 5: the mid-size-batch kernel takes those shapes, K steps in order)


@pytest.mark.parametrize("M,N,K", [(1024, 4096, 4096), (1000, 4096, 1152), (700, 3968, 2176), (1024, 4096, 1024), (130, 16384, 1280)])
def test_gemm_f6_two_k_group_kernel_bit_exact(M, N, K):
    """Shapes with at most one 128x128 tile per CU run the two-K-group kernel with the q kernel's K step (gemm_w4a4_f6qk_kernel:
    odd / even numbers of int4 groups, the shortest K it takes, ragged tiles in both dimensions): the whole output bit for bit
    against the C restatement with the K steps summed in two ordered ranges (atom_gemm_w4a4_f6_order == 2)."""
    from tests import c_oracle
    from tests.helpers import f6_codes
    ops = _ops()
    assert ops.L.lib().atom_gemm_w4a4_f6_order(M, N, K) == 2
    d = rand_gemm_operands(M, N, K, seed=77)
    t = to_device(d, "plain")
    a6 = torch.from_numpy(f6_codes(d["qa4"], d["sA"])).cuda()
    b6 = ops.repack_weight_f6(t[1], t[3])
    out = ops.dense_layer_gemm_i4_fp16(a6, b6, *t[2:], scale_layout="plain", a_wide="f6")
    assert_gemm_close(t2n(out), _exact(d), f"qk {M}x{N}x{K}")
    rows = np.unique(np.r_[0:M:7, M - 70:M])
    want = c_oracle.gemm(O.pack_int4(d["qa4"][rows]), O.pack_int4(d["qb4"]), np.ascontiguousarray(d["sA"][rows].T), d["sB"],
                         d["qa8"][rows], d["qb8"], d["sA8"][rows], d["sB8"], nsplit=-2)
    assert np.array_equal(bits16(t2n(out)[rows]), bits16(want))
    if ops.L.lib().atom_gemm_w4a4_ws_recodes(M, N, K):                        # the reference's operand format: re-coded, same kernel
        assert torch.equal(ops.dense_layer_gemm_i4_fp16(*t, scale_layout="plain"), out)


@pytest.mark.parametrize("M,N,K", [(2048, 4096, 1152), (1500, 4096, 4096), (2048, 3968, 896), (520, 13824, 1280), (1030, 8192, 2176),
                                   (2048, 4096, 640)])
def test_gemm_f6_256x128_kernel_bit_exact(M, N, K):
    """Shapes with more 128x128 tiles than CUs but at most 256 tiles of 256x128 run the 256x128 q-step kernel when the weight carries
    its fp32 scales (gemm_w4a4_f6q2_kernel; ragged tiles in both dimensions, 4..33 int4 groups): the K steps in order -- bit for bit
    against the C restatement on sampled rows, and the whole output against the same weight without the appended scales (the
    128x128 geometry, same order)."""
    from tests import c_oracle
    from tests.helpers import f6_codes
    ops = _ops()
    assert ops.L.lib().atom_gemm_w4a4_f6_order(M, N, K) == 1
    d = rand_gemm_operands(M, N, K, seed=78)
    t = to_device(d, "plain")
    a6 = torch.from_numpy(f6_codes(d["qa4"], d["sA"])).cuda()
    out = ops.dense_layer_gemm_i4_fp16(a6, ops.repack_weight_f6(t[1], t[3]), *t[2:], scale_layout="plain", a_wide="f6")
    assert_gemm_close(t2n(out), _exact(d), f"q2 {M}x{N}x{K}")
    rows = np.unique(np.r_[0:M:11, M - 70:M])
    want = c_oracle.gemm(O.pack_int4(d["qa4"][rows]), O.pack_int4(d["qb4"]), np.ascontiguousarray(d["sA"][rows].T), d["sB"],
                         d["qa8"][rows], d["qb8"], d["sA8"][rows], d["sB8"])
    assert np.array_equal(bits16(t2n(out)[rows]), bits16(want))
    plain = ops.dense_layer_gemm_i4_fp16(a6, ops.repack_weight_f6(t[1]), *t[2:], scale_layout="plain", a_wide="f6")
    assert torch.equal(plain, out)


def test_packed_route_weight_f6_cache():
    """Packed (reference-format) operands of prefill size: atom_amd.ops keeps the weight's re-coded F6 form per WEIGHT (least recently
    used out under a byte cap) and re-codes only the activation on later calls.  Results must not depend on it: repeat calls, another
    batch size, other weights in between (a 7-projection layer walks them in turn: every one must hit from the second pass on),
    eviction, a split-K / o4 / decode user of the scratch workspace in between, and a weight rewritten in place all give the bits of the
    uncached workspace route."""
    from atom_amd import ops
    lib = ops.L.lib()
    N, K = 2048, 1152
    d1 = rand_gemm_operands(1024, N, K, seed=11)
    d2 = rand_gemm_operands(1024, N, K, seed=12)
    assert lib.atom_gemm_w4a4_ws_recodes(1024, N, K) == 1 and lib.atom_gemm_w4a4_ws_recodes(256, N, K) == 0
    dev1, dev2 = to_device(d1, "plain"), to_device(d2, "plain")
    st = ops._F6W_STATE

    def uncached(dev):
        ops.set_weight_f6_cache_bytes(0)
        try:
            return ops.dense_layer_gemm_i4_fp16(*dev, scale_layout="plain")
        finally:
            ops.set_weight_f6_cache_bytes(32 << 30)
    want1, want2 = uncached(dev1), uncached(dev2)
    ops.clear_weight_f6_cache()
    assert torch.equal(ops.dense_layer_gemm_i4_fp16(*dev1, scale_layout="plain"), want1) and (st["hits"], st["misses"]) == (0, 1)
    assert torch.equal(ops.dense_layer_gemm_i4_fp16(*dev1, scale_layout="plain"), want1) and (st["hits"], st["misses"]) == (1, 1)
    # same weight, other activations and another batch size
    mixed = [dev2[0][:800], dev1[1], dev2[2][:, :800].contiguous(), dev1[3], dev2[4][:800], dev1[5], dev2[6][:800], dev1[7]]
    y = ops.dense_layer_gemm_i4_fp16(*mixed, scale_layout="plain")
    assert st["hits"] == 2 and torch.equal(y, uncached(mixed))
    # several weights in turn, twice: the second pass hits every one of them (what a layer's projections do to the cache)
    more = [to_device(rand_gemm_operands(512, N, K, seed=20 + i), "plain") for i in range(5)]
    wants = [uncached(d) for d in more]
    for rep in range(2):
        h0 = st["hits"]
        for d, w in zip(more, wants):
            assert torch.equal(ops.dense_layer_gemm_i4_fp16(*d, scale_layout="plain"), w)
        assert st["hits"] - h0 == (0 if rep == 0 else 5)
    # users of the scratch workspace in between (split-K partial sums, the decode-batch o4 route): nothing of the cache lives there
    assert lib.atom_gemm_w4a4_workspace_bytes(136, 512, 4224) > 0 and lib.atom_gemm_w4a4_ws_recodes(136, 512, 4224) == 0       # split-K route
    sk = to_device(rand_gemm_operands(136, 512, 4224, seed=13), "plain")
    ops.dense_layer_gemm_i4_fp16(*sk, scale_layout="plain")
    ops.dense_layer_gemm_i4_o4(*[t[:16] if i in (0, 4, 6) else (t[:, :16].contiguous() if i == 2 else t) for i, t in enumerate(dev2)], scale_layout="plain")
    assert torch.equal(ops.dense_layer_gemm_i4_fp16(*dev1, scale_layout="plain"), want1)
    # eviction under a byte cap that holds two weights: results unchanged, the footprint stays under the cap
    one = ops._F6W[next(iter(ops._F6W))][2]
    ops.set_weight_f6_cache_bytes(2 * one)
    assert len(ops._F6W) == 2 and st["bytes"] == 2 * one
    for d, w in zip(more, wants):
        assert torch.equal(ops.dense_layer_gemm_i4_fp16(*d, scale_layout="plain"), w) and st["bytes"] <= 2 * one
    ops.set_weight_f6_cache_bytes(32 << 30)
    # the weight rewritten IN PLACE: the version counter changes, the entry does not match
    assert torch.equal(ops.dense_layer_gemm_i4_fp16(*dev1, scale_layout="plain"), want1)
    dev1[1].copy_(dev2[1]); dev1[3].copy_(dev2[3]); dev1[5].copy_(dev2[5]); dev1[7].copy_(dev2[7])
    got = ops.dense_layer_gemm_i4_fp16(*dev1, scale_layout="plain")
    assert torch.equal(got, uncached(dev1))
    # ... and through an alias that keeps the counter: forget_weight_f6 is the documented way
    dev1[1].data.copy_(dev2[1].flip(0)); dev1[3].data.copy_(dev2[3])
    ops.forget_weight_f6(dev1[1])
    assert torch.equal(ops.dense_layer_gemm_i4_fp16(*dev1, scale_layout="plain"), uncached(dev1))
    ops.clear_weight_f6_cache()


def test_weight_f6_cache_lifetime_streams_and_inference_tensors():
    """The per-weight BF6 cache of atom_amd.ops: an entry dies with its packed weight (weakref finalizers on the storages:
    deleting a model frees its BF6 forms, and a freed address cannot hit a stale entry); an entry made on one stream is complete for a
    call on another; weights created under torch.inference_mode() (no version counter) take the cache too."""
    import gc
    from atom_amd import ops
    N, K = 2048, 1152
    ops.clear_weight_f6_cache()
    st = ops._F6W_STATE
    d = to_device(rand_gemm_operands(512, N, K, seed=31), "plain")
    ops.set_weight_f6_cache_bytes(0)
    want = ops.dense_layer_gemm_i4_fp16(*d, scale_layout="plain")
    ops.set_weight_f6_cache_bytes(32 << 30)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                            # the entry is made on a side stream ...
        y0 = ops.dense_layer_gemm_i4_fp16(*d, scale_layout="plain")
    assert len(ops._F6W) == 1 and st["misses"] == 1
    y1 = ops.dense_layer_gemm_i4_fp16(*d, scale_layout="plain")   # ... and hit from the default stream straight away
    side.synchronize()
    assert st["hits"] == 1 and torch.equal(y0, want) and torch.equal(y1, want)
    # the weight goes away: so does the entry
    d = list(d)
    d[1] = d[3] = None
    gc.collect()
    assert len(ops._F6W) == 0 and st["bytes"] == 0, (len(ops._F6W), st["bytes"])
    # inference tensors
    with torch.inference_mode():
        di = to_device(rand_gemm_operands(512, N, K, seed=31), "plain")
        y2 = ops.dense_layer_gemm_i4_fp16(*di, scale_layout="plain")
        y3 = ops.dense_layer_gemm_i4_fp16(*di, scale_layout="plain")
    assert torch.equal(y2, want) and torch.equal(y3, want) and len(ops._F6W) == 1
    del di
    gc.collect()
    ops.clear_weight_f6_cache()


def test_packed_route_under_hip_graph_capture():
    """Round-3 advisor / verdict item: a cached weight must never make a captured graph depend on what other calls left behind.
    (a) capture with the weight NOT cached: nothing executes during capture, so no entry may be created -- the captured call re-codes
    both operands inside the graph; (b) capture with the weight cached: the captured launch points at the entry, which is pinned.
    In both cases: capture(weight A) -> eager calls with weight B (and eviction pressure) -> replay == the uncached result, and an
    eager call right after a capture is right too."""
    from atom_amd import ops
    N, K = 2048, 1152
    A = to_device(rand_gemm_operands(512, N, K, seed=31), "plain")
    B = to_device(rand_gemm_operands(512, N, K, seed=32), "plain")
    ops.set_weight_f6_cache_bytes(0)
    wantA = ops.dense_layer_gemm_i4_fp16(*A, scale_layout="plain")
    wantB = ops.dense_layer_gemm_i4_fp16(*B, scale_layout="plain")
    ops.set_weight_f6_cache_bytes(32 << 30)
    for precached in (False, True):
        ops.clear_weight_f6_cache()
        if precached:
            ops.dense_layer_gemm_i4_fp16(*A, scale_layout="plain")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = ops.dense_layer_gemm_i4_fp16(*A, scale_layout="plain")
        entries = list(ops._F6W.values())
        assert len(entries) == (1 if precached else 0) and all(e[3] for e in entries)         # created never, pinned if hit
        # an eager call right after the capture (the captured launches have not run yet)
        assert torch.equal(ops.dense_layer_gemm_i4_fp16(*A, scale_layout="plain"), wantA)
        # another weight on the same stream / scratch workspace, then eviction pressure
        assert torch.equal(ops.dense_layer_gemm_i4_fp16(*B, scale_layout="plain"), wantB)
        ops.set_weight_f6_cache_bytes(1)
        ops.set_weight_f6_cache_bytes(32 << 30)
        assert len(ops._F6W) == (1 if precached else 0)                                       # the pinned entry survives
        y.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, wantA)
        # a second graph captured on the same capture stream, starting with the same weight (the advisor's scenario)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            y2 = ops.dense_layer_gemm_i4_fp16(*A, scale_layout="plain")
        ops.dense_layer_gemm_i4_fp16(*B, scale_layout="plain")
        g2.replay()
        torch.cuda.synchronize()
        assert torch.equal(y2, wantA)
        del g, g2
    ops.clear_weight_f6_cache()


def test_c_abi_ws_weight_cached_flag():
    """ATOM_WS_WEIGHT_CACHED of the C ABI (include/atom_hip.h; the contract a foreign caller that owns its workspace gets): the second
    call with the same weight and workspace may assert the flag and gets the bits of the first."""
    from atom_amd import ops
    L = ops.L
    lib = L.lib()
    M, N, K = 512, 2048, 1152
    dev = to_device(rand_gemm_operands(M, N, K, seed=41), "plain")
    a, b, sa, sb, a8, b8, sa8, sb8 = dev
    need = lib.atom_gemm_w4a4_workspace_bytes(M, N, K)
    assert need > 0 and lib.atom_gemm_w4a4_ws_recodes(M, N, K) == 1
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    outs = []
    for flag in (0, L.WS_WEIGHT_CACHED, L.WS_WEIGHT_CACHED):
        d = torch.empty((M, N), dtype=torch.float16, device="cuda")
        st = lib.atom_gemm_w4a4_f16_ws(a.data_ptr(), b.data_ptr(), sa.data_ptr(), sb.data_ptr(), a8.data_ptr(), b8.data_ptr(), sa8.data_ptr(),
                                       sb8.data_ptr(), d.data_ptr(), M, N, K, 128, 128, L.SCALE_LAYOUT_PLAIN | flag, ws.data_ptr(), need,
                                       L.current_stream(a.device))
        L.check(st, "atom_gemm_w4a4_f16_ws")
        outs.append(d)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(outs[0], ops.dense_layer_gemm_i4_fp16(*dev, scale_layout="plain"))


@pytest.mark.parametrize("M,N,K", [(64, 13824, 5120), (200, 4096, 4224), (40, 5120, 13824), (512, 4096, 1152), (300, 2048, 1152)])
def test_c_abi_one_workspace_per_weight_recipe(M, N, K):
    """The recipe of INTEGRATION.md for a binding that owns its layers: the weight region of the workspace filled ONCE by
    atom_repack_weight_f6s, ATOM_WS_WEIGHT_CACHED in every call -- the library then re-codes only the activation wherever
    atom_gemm_w4a4_ws_recodes_cached says so (from 129 rows, and from 17 where the decode-batch kernel does not take the shape) and the
    result is the plain entry point's in the order atom_gemm_w4a4_packed_order(.., 2) names -- here always the K steps in order, hence
    bit-identical to the call without a workspace where that sums in order too."""
    from atom_amd import ops
    L = ops.L
    lib = L.lib()
    dev = to_device(rand_gemm_operands(M, N, K, seed=M + N + K), "plain")
    a, b, sa, sb, a8, b8, sa8, sb8 = dev
    need = lib.atom_gemm_w4a4_workspace_bytes(M, N, K)
    assert lib.atom_gemm_w4a4_ws_recodes_cached(M, N, K) == 1 and need >= lib.atom_f6_weight_bytes(N, K)
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    stream = L.current_stream(a.device)
    L.check(lib.atom_repack_weight_f6s(b.data_ptr(), sb.data_ptr(), N, K, ws.data_ptr(), stream), "atom_repack_weight_f6s")
    pairs = L.B_SCALE_PAIRS if ops.scale_pairs_shared(sb, N) else 0
    flags = L.SCALE_LAYOUT_PLAIN | pairs | L.WS_WEIGHT_CACHED
    outs = []
    for _ in range(2):
        d = torch.empty((M, N), dtype=torch.float16, device="cuda")
        L.check(lib.atom_gemm_w4a4_f16_ws(a.data_ptr(), b.data_ptr(), sa.data_ptr(), sb.data_ptr(), a8.data_ptr(), b8.data_ptr(), sa8.data_ptr(),
                                          sb8.data_ptr(), d.data_ptr(), M, N, K, 128, 128, flags, ws.data_ptr(), need, stream), "atom_gemm_w4a4_f16_ws")
        outs.append(d)
    assert torch.equal(outs[0], outs[1])
    assert lib.atom_gemm_w4a4_packed_order(M, N, K, 2) == 1
    plain = torch.empty_like(outs[0])
    L.check(lib.atom_gemm_w4a4_f16(a.data_ptr(), b.data_ptr(), sa.data_ptr(), sb.data_ptr(), a8.data_ptr(), b8.data_ptr(), sa8.data_ptr(),
                                   sb8.data_ptr(), plain.data_ptr(), M, N, K, 128, 128, L.SCALE_LAYOUT_PLAIN | pairs, stream), "atom_gemm_w4a4_f16")
    if lib.atom_gemm_w4a4_packed_order(M, N, K, 0) == 1:
        assert torch.equal(outs[0], plain)
    assert_gemm_close(t2n(outs[0]), t2n(plain).astype(np.float64), f"cached-weight route {M}x{N}x{K}")


def _ws_call(lib, L, dev, M, N, K, flags, ws, need):
    a, b, sa, sb, a8, b8, sa8, sb8 = dev
    d = torch.empty((M, N), dtype=torch.float16, device="cuda")
    st = lib.atom_gemm_w4a4_f16_ws(a.data_ptr(), b.data_ptr(), sa.data_ptr(), sb.data_ptr(), a8.data_ptr(), b8.data_ptr(), sa8.data_ptr(),
                                   sb8.data_ptr(), d.data_ptr(), M, N, K, 128, 128, flags, ws.data_ptr(), need, L.current_stream(a.device))
    return st, d


def test_c_abi_cached_weight_survives_decode_size_calls():
    """one workspace per weight, ATOM_WS_WEIGHT_CACHED on EVERY call, batch sizes alternating between 8 .. 16 rows
    and >= 17 rows at K_total > 14464 (Llama-70B down_proj's K = 28672 at a reduced N): the decode-size call used to run split-K with
    its FP32 partial sums at workspace offset 0 -- over the cached BF6 weight -- and the next call from 17 rows multiplied garbage.  A
    call with the flag outside the re-coding route now runs as the plain entry point does (atom_gemm_w4a4_packed_order(.., 2))."""
    from atom_amd import ops
    L = ops.L
    lib = L.lib()
    N, K = 2048, 14464 + 128 * 4
    need = max(lib.atom_gemm_w4a4_workspace_bytes(M, N, K) for M in (8, 12, 16, 24, 160))
    assert need >= lib.atom_f6_weight_bytes(N, K)
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    full = rand_gemm_operands(160, N, K, seed=77)
    stream = L.current_stream(torch.device("cuda"))
    devs = {}
    for M in (8, 12, 16, 24, 160):
        sub = dict(full)
        for k in ("qa4", "qa8", "sA", "sA8"):
            sub[k] = full[k][:M]
        devs[M] = to_device(sub, "plain")
    b, sb = devs[160][1], devs[160][3]
    L.check(lib.atom_repack_weight_f6s(b.data_ptr(), sb.data_ptr(), N, K, ws.data_ptr(), stream), "atom_repack_weight_f6s")
    region = ws[: lib.atom_f6_weight_bytes(N, K)].clone()
    flags = L.SCALE_LAYOUT_PLAIN | L.B_SCALE_PAIRS | L.WS_WEIGHT_CACHED
    # the workspace-less split-K shape the finding names: without the flag it does split (order 100 + s), with it it must not
    assert lib.atom_gemm_w4a4_packed_order(12, N, K, 1) > 100 and lib.atom_gemm_w4a4_packed_order(12, N, K, 2) < 100
    ref = {}
    for M in (24, 8, 160, 12, 24, 16, 160):
        st, d = _ws_call(lib, L, devs[M], M, N, K, flags, ws, need)
        L.check(st, "atom_gemm_w4a4_f16_ws")
        torch.cuda.synchronize()
        assert torch.equal(ws[: region.numel()], region), f"M = {M} wrote into the cached weight region"
        if M in ref:
            assert torch.equal(d, ref[M]), f"M = {M}: result changed after decode-size calls on the same workspace"
        ref[M] = d
        want = torch.empty_like(d)
        a, b_, sa, sb_, a8, b8, sa8, sb8 = devs[M]
        L.check(lib.atom_gemm_w4a4_f16(a.data_ptr(), b_.data_ptr(), sa.data_ptr(), sb_.data_ptr(), a8.data_ptr(), b8.data_ptr(), sa8.data_ptr(),
                                       sb8.data_ptr(), want.data_ptr(), M, N, K, 128, 128, L.SCALE_LAYOUT_PLAIN | L.B_SCALE_PAIRS, stream), "plain")
        assert_gemm_close(t2n(d), t2n(want).astype(np.float64), f"cached-weight workspace, M = {M}")


def test_c_abi_wrong_assertions_are_caught_in_verify_mode():
    """ATOM_B_SCALE_PAIRS and ATOM_WS_WEIGHT_CACHED are caller assertions whose violation gives wrong numbers
    silently.  atom_check_scale_pairs / atom_verify_weight_f6s count the violations on the device, and a debug call with ATOM_WS_VERIFY
    returns ATOM_ERR_INVALID_ARG instead of computing with a wrong assertion (the reference checks at its binding too: TORCH_CHECK in
    punica_ops.cc:18-47)."""
    from atom_amd import ops
    L = ops.L
    lib = L.lib()
    M, N, K = 300, 2048, 1152
    ops_np = rand_gemm_operands(M, N, K, seed=5)
    dev = to_device(ops_np, "plain")
    a, b, sa, sb, a8, b8, sa8, sb8 = dev
    stream = L.current_stream(a.device)
    G = K // 128 - 1
    cnt = torch.full((4,), -1, dtype=torch.int32, device="cuda")
    L.check(lib.atom_check_scale_pairs(sb.data_ptr(), G, N, cnt.data_ptr(), stream), "atom_check_scale_pairs")
    assert int(cnt[0]) == 0
    sb_bad = sb.clone()
    sb_bad[3, 11] = sb_bad[3, 10] * 2                                    # one pair with two different scales
    sb_bad[0, 1] = sb_bad[0, 0] + sb_bad[0, 0] * 0.5
    L.check(lib.atom_check_scale_pairs(sb_bad.data_ptr(), G, N, cnt.data_ptr(), stream), "atom_check_scale_pairs")
    assert int(cnt[0]) == 2
    need = lib.atom_gemm_w4a4_workspace_bytes(M, N, K)
    assert need > 0 and lib.atom_gemm_w4a4_ws_recodes_cached(M, N, K) == 1
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    L.check(lib.atom_repack_weight_f6s(b.data_ptr(), sb.data_ptr(), N, K, ws.data_ptr(), stream), "atom_repack_weight_f6s")
    L.check(lib.atom_verify_weight_f6s(b.data_ptr(), sb.data_ptr(), N, K, ws.data_ptr(), cnt.data_ptr(), stream), "atom_verify_weight_f6s")
    assert int(cnt[0]) == 0
    base = L.SCALE_LAYOUT_PLAIN | L.WS_VERIFY
    # right assertions: the verified call computes what the unverified one does
    st, d_ok = _ws_call(lib, L, dev, M, N, K, base | L.B_SCALE_PAIRS | L.WS_WEIGHT_CACHED, ws, need)
    L.check(st, "verified call")
    st, d_plain = _ws_call(lib, L, dev, M, N, K, L.SCALE_LAYOUT_PLAIN | L.B_SCALE_PAIRS | L.WS_WEIGHT_CACHED, ws, need)
    L.check(st, "unverified call")
    assert torch.equal(d_ok, d_plain)
    # wrong ATOM_B_SCALE_PAIRS: the scales do not pair up
    dev_bad = (a, b, sa, sb_bad, a8, b8, sa8, sb8)
    st, _ = _ws_call(lib, L, dev_bad, M, N, K, base | L.B_SCALE_PAIRS, ws, need)
    assert st == L.ERR_INVALID_ARG, lib.atom_strerror(st).decode()
    st, _ = _ws_call(lib, L, dev_bad, M, N, K, base, ws, need)            # without the assertion the same operands are fine
    L.check(st, "unpaired scales, no assertion")
    # wrong ATOM_WS_WEIGHT_CACHED: the workspace holds ANOTHER weight's BF6 form (one code changed) / a zeroed region
    L.check(lib.atom_repack_weight_f6s(b.data_ptr(), sb.data_ptr(), N, K, ws.data_ptr(), stream), "atom_repack_weight_f6s")
    b2 = b.clone()
    b2[17, 5] ^= 0x30
    L.check(lib.atom_verify_weight_f6s(b2.data_ptr(), sb.data_ptr(), N, K, ws.data_ptr(), cnt.data_ptr(), stream), "atom_verify_weight_f6s")
    assert int(cnt[0]) == 1
    dev2 = (a, b2, sa, sb, a8, b8, sa8, sb8)
    st, _ = _ws_call(lib, L, dev2, M, N, K, base | L.WS_WEIGHT_CACHED, ws, need)
    assert st == L.ERR_INVALID_ARG, lib.atom_strerror(st).decode()
    ws.zero_()
    st, _ = _ws_call(lib, L, dev, M, N, K, base | L.WS_WEIGHT_CACHED, ws, need)
    assert st == L.ERR_INVALID_ARG, lib.atom_strerror(st).decode()
    st, d2 = _ws_call(lib, L, dev, M, N, K, base, ws, need)                # no assertion: both operands re-coded, right answer
    L.check(st, "no assertion")
    assert torch.equal(d2, d_ok)


@pytest.mark.parametrize("M,N,K", [(1, 256, 256), (1, 4096, 4096), (1, 11008, 4096), (1, 1024, 11008), (1, 13824, 5120), (1, 640, 13824),
                                   (1, 8256, 640), (1, 8320, 384), (2, 1024, 11008), (2, 13824, 5120), (2, 640, 13824), (2, 8320, 4224)])
@pytest.mark.parametrize("layout", ["plain", "ref"])
def test_gemv_bit_exact_vs_c_contract(M, N, K, layout):
    """The dot-product kernel (BASELINE config 2: one token; two tokens where K > 4096; round 4: one 16-byte chunk per feature and K
    batch, one or two adjacent features per wave from 8192 features up, XCD-aware feature map -- csrc/gemv_w4a4.hip) is bit-identical
    to the C restatement of its summation order (oracle/atom_oracle.c oracle_gemv_w4a4_f16_lanes: groups dealt to the 16 quad leaders
    in ascending order, 64-lane butterfly, keeper last; every token independently) -- the order the round-3 one-token kernel had, so
    the rewrite changed no bit; short rows (8 chunks: most lanes re-read the last chunk) and both feature-per-wave variants included."""
    from tests import c_oracle as C
    ops = _ops()
    d = rand_gemm_operands(M, N, K, seed=N + K + M)
    dev = to_device(d, layout)
    want = C.gemm(O.pack_int4(d["qa4"]), O.pack_int4(d["qb4"]), d["sA"].T, d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"], nsplit="lanes")
    out = ops.dense_layer_gemm_i4_fp16(*dev, scale_layout=layout)
    assert np.array_equal(bits16(t2n(out)), bits16(want))
    # the FP32 sums and the segmented entry point take the same kernel: same sums, rounded once
    f32 = ops.dense_layer_gemm_i4_f32(*dev, scale_layout=layout)
    assert torch.equal(f32.half(), out)


def test_two_tokens_take_the_decode_batch_kernel_up_to_k_4096():
    """The routing rule is (M, K) only: two tokens with K <= 4096 keep the MFMA decode-batch kernel (its order: nsplit = 8); beyond, and
    for one token, the dot-product kernel (round 6 measured two tokens on it at every K: slower, see csrc/gemm_w4a4.hip gemv_tokens)."""
    from tests import c_oracle as C
    ops = _ops()
    lib = ops.L.lib()
    assert lib.atom_gemm_w4a4_packed_order(2, 1024, 4096, 0) == 8 and lib.atom_gemm_w4a4_packed_order(2, 1024, 4224, 0) == 64
    d = rand_gemm_operands(2, 1024, 4096, seed=77)
    out = ops.dense_layer_gemm_i4_fp16(*to_device(d, "plain"), scale_layout="plain")
    want = C.gemm(O.pack_int4(d["qa4"]), O.pack_int4(d["qb4"]), d["sA"].T, d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"], nsplit=8)
    assert np.array_equal(bits16(t2n(out)), bits16(want))


# mid-size batches in the packed format: gemm_w4a4_mid.hip, INT8 form (64 x 64 tiles over the whole K range on a deep LDS ring).  The
# dispatch gives it what the decode-batch kernel does not take and one tile per CU covers: up to 256 rows, 96 .. 256 tiles, K <= 11264
# (gemm_w4a4.hip mid_fits).  Ragged token tiles, 33 .. 86 int4 groups, both scale layouts, pair-shared and per-channel weight scales.
MID = [(64, 13824, 5120), (40, 13824, 4224), (100, 8192, 8192), (256, 4096, 11008), (129, 4096, 6272), (77, 6208, 10368)]


@pytest.mark.parametrize("M,N,K", MID)
@pytest.mark.parametrize("layout", ["ref", "plain"])
def test_gemm_mid_batches_bit_exact_vs_c_contract(M, N, K, layout):
    """The mid-size-batch kernel sums the K steps in order -- the contract of include/atom_hip.h as oracle/atom_oracle.c restates it
    (oracle_gemm_w4a4_f16) -- so outputs are compared bit for bit on sampled rows and columns (the C loop is O(M N K)).  Called through
    the plain C entry point (no workspace: the INT8 form) and through atom_amd.ops (which, above 128 rows, re-codes the activation and
    runs the BF6 form on its cached weight: the same order here)."""
    from tests import c_oracle as C
    ops = _ops()
    lib = ops.L.lib()
    assert lib.atom_gemm_w4a4_packed_order(M, N, K, 0) == 1 and lib.atom_gemm_w4a4_packed_order(M, N, K, 2) == 1
    d = rand_gemm_operands(M, N, K, seed=M * 3 + N + K)
    if (M + N) % 2:                                          # per-channel weight scales: the kernels without the shared products
        g = np.random.default_rng(M)
        d["sB"] = g.uniform(0.005, 0.05, size=d["sB"].shape).astype(np.float16)
    t = to_device(d, layout)
    out = ops.dense_layer_gemm_i4_fp16(*t, scale_layout=layout)
    assert torch.equal(out, ops.dense_layer_gemm_i4_fp16(*t, scale_layout=layout))
    plain = torch.empty_like(out)
    flags = ops._LAYOUTS[layout] | (ops.L.B_SCALE_PAIRS if ops.scale_pairs_shared(t[3], N) else 0)
    ops.L.check(lib.atom_gemm_w4a4_f16(*[x.data_ptr() for x in t], plain.data_ptr(), M, N, K, 128, 128, flags,
                                       ops.L.current_stream(plain.device)), "atom_gemm_w4a4_f16")
    assert torch.equal(plain, out)                           # INT8 form == BF6 form / ops route: one order, exact integer dots
    g = np.random.default_rng(N)
    rows = np.unique(np.concatenate([[0, M - 1, min(63, M - 1), min(64, M - 1)], g.integers(0, M, 12)]))
    cols = np.unique(np.concatenate([np.arange(0, 72), [N - 1, N - 64, N - 33], g.integers(0, N, 100)]))
    want = C.gemm(O.pack_int4(d["qa4"][rows]), O.pack_int4(d["qb4"][cols]), d["sA"][rows].T, d["sB"][:, cols], d["qa8"][rows],
                  d["qb8"][cols], d["sA8"][rows], d["sB8"][cols])
    got = t2n(plain)[np.ix_(rows, cols)]
    assert np.array_equal(bits16(got), bits16(want)), f"{(bits16(got) != bits16(want)).sum()} of {got.size} sampled elements differ"
    exact = gemm_ref_torch_f64(d).cpu().numpy()
    assert_gemm_close(t2n(out), exact, f"mid batch {M}x{N}x{K} {layout}")
