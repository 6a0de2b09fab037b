"""GPU parity tests of the three fused activation-quant ops and the weight packer, through the C ABI
(atom_amd.ops -> ctypes -> libatom_hip.so), against the CPU oracle and the reference-generated goldens."""
import os

import numpy as np
import pytest
import torch

from oracle import atom_oracle as O
from tests.helpers import bits16, rand_act, scales_plain, t2n, wide_codes

pytestmark = pytest.mark.gpu


def _ops():
    from atom_amd import ops
    return ops


def _check_tail(outs, ref, M, layout, exact=True, xq_ref=None):
    o8, o4, s8, s4 = outs[:4]
    q4 = O.unpack_int4(t2n(o4).view(np.uint8))
    q8 = t2n(o8)
    s4p = scales_plain(s4, M, layout).T          # -> [M, G]
    s8p = scales_plain(s8, M, layout)
    if exact:
        assert np.array_equal(q4, ref["q4"]), f"int4 codes differ at {np.argwhere(q4 != ref['q4'])[:5]}"
        assert np.array_equal(q8, ref["q8"])
        assert np.array_equal(bits16(s4p), bits16(ref["s4"]))
        assert np.array_equal(bits16(s8p), bits16(ref["s8"]))
        if len(outs) == 5 and xq_ref is not None:
            assert np.array_equal(bits16(t2n(outs[4])), bits16(xq_ref))
    else:
        # transcendental (expf) differs by ulps between libraries: codes +-1 on a tiny fraction, scales 1 fp16 ulp
        d4 = np.abs(q4.astype(np.int32) - ref["q4"])
        d8 = np.abs(q8.astype(np.int32) - ref["q8"])
        assert d4.max() <= 1 and d8.max() <= 1
        assert (d4 > 0).mean() < 2e-3 and (d8 > 0).mean() < 2e-3
        assert np.abs(bits16(s4p).astype(np.int32) - bits16(ref["s4"]).astype(np.int32)).max() <= 1
        assert np.abs(bits16(s8p).astype(np.int32) - bits16(ref["s8"]).astype(np.int32)).max() <= 1


SHAPES = [(1, 256), (9, 1024), (33, 4096), (5, 5120), (3, 11008), (2, 13824), (2, 16384), (300, 512)]


@pytest.mark.parametrize("M,H", SHAPES)
@pytest.mark.parametrize("mode,clip", [("sim", 0.9), ("kernel", 1.0), ("sim", 1.0), ("kernel", 0.9)])
@pytest.mark.parametrize("layout", ["ref", "plain"])
def test_reorder_quant_bit_exact(M, H, mode, clip, layout):
    ops = _ops()
    x = rand_act(M, H, seed=M * 131 + H)
    idx = np.random.default_rng(H).permutation(H).astype(np.int16)
    ref = O.reorder_quant(x, idx, mode, clip)
    outs = ops.reorder_fp16_i4(torch.from_numpy(x).cuda(), torch.from_numpy(idx).cuda(), quant_mode=mode,
                               clip=clip, scale_layout=layout, return_dequant=True)
    xq_ref = O.act_dequant_sim(ref) if mode == "sim" else None
    _check_tail(outs, ref, M, layout, exact=True, xq_ref=xq_ref)


def test_reorder_quant_identity_and_edges():
    ops = _ops()
    M, H = 6, 1024
    x = rand_act(M, H, seed=5)
    x[1] = 0                                   # all-zero row: amax clamp (sim) / zero scale (kernel)
    x[2, :] = 1e-6                             # below the 1e-5 clamp
    x[3, 7] = 65504.0                          # fp16 max
    x[4] = (np.arange(H) % 15 - 7) * 0.5       # exact ties
    for mode, clip in [("sim", 0.9), ("kernel", 1.0)]:
        ref = O._quant_row_tail(x if mode == "sim" else x.astype(np.float32), mode, clip)
        outs = ops.reorder_fp16_i4(torch.from_numpy(x).cuda(), None, quant_mode=mode, clip=clip,
                                   scale_layout="plain", return_dequant=True)
        _check_tail(outs, ref, M, "plain", exact=True,
                    xq_ref=O.act_dequant_sim(ref) if mode == "sim" else None)


def test_reorder_quant_golden(golden_dir):
    """Against the reference's own output (index_select + quantize_activation_wrapper)."""
    ops = _ops()
    z = np.load(os.path.join(golden_dir, "reorder_quant_9x1024.npz"))
    outs = ops.reorder_fp16_i4(torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["idx"]).cuda(),
                               quant_mode="sim", clip=0.9, scale_layout="plain", return_dequant=True)
    assert np.array_equal(bits16(t2n(outs[4])), bits16(z["xq"]))


@pytest.mark.parametrize("clip", [1.0 / 64.0, 0.5, 0.0155])
def test_simulated_path_small_clips_and_edge_values(clip):
    """The simulated path runs in the FP16 domain (quant_math.h sim_codes4) for clip >= 1/64 and in the FP32 form below (a smaller
    clip could overflow the half estimate of the quotient): both against the oracle, bit for bit, on rows that hold every edge the
    arithmetic has -- all-zero groups (the 1e-5 clamp makes the scale a subnormal half), values far below the scale, subnormal
    halves, fp16 max, exact ties of the half quotient, -0 -- through all three quantisers and the de-quantised output."""
    ops = _ops()
    M, H = 12, 1024
    x = rand_act(M, H, seed=11)
    x[1] = 0
    x[2, :] = 1e-6
    x[3, 7] = 65504.0
    x[4] = (np.arange(H) % 15 - 7) * 0.5
    x[5] = np.float16(6e-8) * (np.arange(H) % 5)            # subnormal halves
    x[6, ::2] = -0.0
    x[7] = (np.arange(H) % 17 - 8) * np.float16(0.000123)
    x[8, :128] = 0                                          # one all-zero group in an ordinary row
    x[9] *= np.float16(1e-3)
    x[10] = np.where(np.arange(H) % 128 == 0, np.float16(30000.0), x[10] * np.float16(1e-4))   # huge group maximum, tiny values
    if clip < 0.1:                                          # clamp(1e-5) * clip / 7 rounds to a ZERO half: the reference divides by 0
        x[[1, 2, 5]] = rand_act(3, H, seed=13) * np.float16(1e-2)       # there (NaN codes); keep these rows' scales representable
        x[8, :128] = x[9, :128]
        x[10] = rand_act(1, H, seed=14)[0] * np.float16(4.0)              # (the normalised row 10 has groups of ~1e-8)
    idx = np.random.default_rng(3).permutation(H).astype(np.int16)
    ref = O.reorder_quant(x, idx, "sim", clip)
    outs = ops.reorder_fp16_i4(torch.from_numpy(x).cuda(), torch.from_numpy(idx).cuda(), quant_mode="sim", clip=clip,
                               scale_layout="plain", return_dequant=True)
    _check_tail(outs, ref, M, "plain", exact=True, xq_ref=O.act_dequant_sim(ref))
    ref = O._quant_row_tail(x, "sim", clip)
    outs = ops.reorder_fp16_i4(torch.from_numpy(x).cuda(), None, quant_mode="sim", clip=clip, scale_layout="plain", return_dequant=True)
    _check_tail(outs, ref, M, "plain", exact=True, xq_ref=O.act_dequant_sim(ref))
    w = (1 + 0.1 * np.random.default_rng(4).standard_normal(H)).astype(np.float16)
    ref = O.rmsnorm_reorder_quant(x, w, 1e-5, idx, "sim", clip)
    outs = ops.rmsnorm_fp16_i4(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), torch.from_numpy(idx).cuda(), 1e-5,
                               quant_mode="sim", clip=clip, scale_layout="plain", return_dequant=True)
    _check_tail(outs, ref, M, "plain", exact=True, xq_ref=O.act_dequant_sim(ref))
    b = rand_act(M, H, seed=12)
    a = x.copy()                                            # (silu(65504) * b overflows the half product: inf scales, NaN codes)
    a[3, 7] = 4.0
    a[10] = x[9]
    ref = O.silu_mul_quant(a, b, "sim", clip)
    outs = ops.activate_fp16_i4(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), quant_mode="sim", clip=clip,
                                scale_layout="plain", return_dequant=True)
    _check_tail(outs, ref, M, "plain", exact=False)


@pytest.mark.parametrize("M,H", [(1, 256), (7, 1024), (33, 4096), (4, 5120), (2, 13824)])
@pytest.mark.parametrize("mode,clip", [("sim", 0.9), ("kernel", 1.0)])
def test_rmsnorm_quant_bit_exact(M, H, mode, clip):
    ops = _ops()
    g = np.random.default_rng(H + M)
    x = (rand_act(M, H, seed=M + 7 * H).astype(np.float32) * 2.0).astype(np.float16)
    w = (1.0 + 0.1 * g.standard_normal(H)).astype(np.float16)
    idx = g.permutation(H).astype(np.int16)
    eps = 1e-5
    ref = O.rmsnorm_reorder_quant(x, w, eps, idx, mode, clip)
    outs = ops.rmsnorm_fp16_i4(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), torch.from_numpy(idx).cuda(),
                               eps, quant_mode=mode, clip=clip, scale_layout="ref", return_dequant=True)
    _check_tail(outs, ref, M, "ref", exact=True, xq_ref=O.act_dequant_sim(ref) if mode == "sim" else None)


def test_rmsnorm_quant_golden(golden_dir):
    """QLlamaRMSNorm.forward of the reference (HF RMSNorm uses torch.rsqrt / mean: last-bit differences in the FP32
    statistics can flip an fp16 rounding) -> the reference's own tolerance: codes +-1, scales 1e-3."""
    ops = _ops()
    z = np.load(os.path.join(golden_dir, "rmsnorm_quant_7x1024.npz"))
    outs = ops.rmsnorm_fp16_i4(torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["w"]).cuda(),
                               torch.from_numpy(z["idx"]).cuda(), float(z["eps"]), quant_mode="sim", clip=0.9,
                               scale_layout="plain", return_dequant=True)
    got = t2n(outs[4]).astype(np.float32)
    want = z["xq"].astype(np.float32)
    mism = bits16(got) != bits16(want)
    assert mism.mean() < 5e-3
    assert np.abs(got - want).max() <= np.abs(want).max() / 7 * 1.01    # at most one INT4 step of the largest group


@pytest.mark.parametrize("M,H", [(1, 256), (5, 1408), (3, 11008), (2, 13824)])
@pytest.mark.parametrize("mode,clip", [("sim", 0.9), ("kernel", 1.0)])
def test_silu_mul_quant(M, H, mode, clip):
    ops = _ops()
    g = np.random.default_rng(M * H)
    a = (g.standard_normal((M, H)) * 2).astype(np.float16)
    b = (g.standard_normal((M, H)) * 2).astype(np.float16)
    ref = O.silu_mul_quant(a, b, mode, clip)
    outs = ops.activate_fp16_i4(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), quant_mode=mode, clip=clip,
                                scale_layout="ref")
    _check_tail(outs, ref, M, "ref", exact=False)


def test_silu_mul_quant_golden(golden_dir):
    ops = _ops()
    z = np.load(os.path.join(golden_dir, "silu_mul_quant_5x1408.npz"))
    outs = ops.activate_fp16_i4(torch.from_numpy(z["a"]).cuda(), torch.from_numpy(z["b"]).cuda(), quant_mode="sim",
                                clip=0.9, scale_layout="plain", return_dequant=True)
    mism = bits16(t2n(outs[4])) != bits16(z["xq"])
    assert mism.mean() < 5e-3


@pytest.mark.parametrize("N,K", [(512, 512), (256, 640), (64, 4096), (128, 11008)])
def test_weight_quant_bit_exact(N, K, golden_dir):
    ops = _ops()
    if (N, K) == (512, 512):
        W = np.load(os.path.join(golden_dir, "c1_qlinear_16x512x512.npz"))["W"]
        Wq_gold = np.load(os.path.join(golden_dir, "c1_qlinear_16x512x512.npz"))["Wq"]
    elif (N, K) == (256, 640):
        W = np.load(os.path.join(golden_dir, "weight_quant_256x640.npz"))["W"]
        Wq_gold = np.load(os.path.join(golden_dir, "weight_quant_256x640.npz"))["Wq"]
    else:
        W = (np.random.default_rng(N + K).standard_normal((N, K)) * 0.05).astype(np.float16)
        Wq_gold = None
    ref = O.quant_weight_sim(W, 0.85, 2)
    b4, b8, sb, sb8, wq = ops.quant_weight_w4(torch.from_numpy(W).cuda(), 0.85, 2, return_fake_quant=True)
    assert np.array_equal(O.unpack_int4(t2n(b4)), ref["q4"])
    assert np.array_equal(t2n(b8), ref["q8"])
    assert np.array_equal(bits16(t2n(sb)), bits16(ref["s4"]))
    assert np.array_equal(bits16(t2n(sb8)), bits16(ref["s8"]))
    assert np.array_equal(bits16(t2n(wq)), bits16(ref["wq"]))
    if Wq_gold is not None:                    # the reference's own QLinearLayer.quant() output
        assert np.array_equal(bits16(t2n(wq)), bits16(Wq_gold))


def test_error_behaviour():
    """The C ABI validates before launching (the reference validates nothing)."""
    ops = _ops()
    from atom_amd._lib import AtomHipError
    with pytest.raises(AtomHipError):
        ops.reorder_fp16_i4(torch.zeros((2, 200), dtype=torch.float16, device="cuda"), None)     # H % 128 != 0
    with pytest.raises(AtomHipError):
        ops.reorder_fp16_i4(torch.zeros((2, 256), dtype=torch.float16), None)                    # CPU tensor
    with pytest.raises(AtomHipError):
        ops.reorder_fp16_i4(torch.zeros((0, 256), dtype=torch.float16, device="cuda"), None)     # empty


@pytest.mark.parametrize("M,H", [(1, 256), (9, 1024), (33, 4096), (3, 11008), (300, 512)])
@pytest.mark.parametrize("mode,clip", [("sim", 0.9), ("kernel", 1.0)])
def test_wide_codes_are_the_packed_codes_widened(M, H, mode, clip):
    """ATOM_QUANT_WIDE_CODES: same codes, stored as the INT8 MFMA operand (x16, even/odd de-interleaved per 32 channels);
    everything else (keeper, scales, de-quantised tensor) identical to the packed call."""
    ops = _ops()
    x = rand_act(M, H, seed=M * 3 + H)
    idx = np.random.default_rng(H).permutation(H).astype(np.int16)
    ref = O.reorder_quant(x, idx, mode, clip)
    xt, it = torch.from_numpy(x).cuda(), torch.from_numpy(idx).cuda()
    p = ops.reorder_fp16_i4(xt, it, quant_mode=mode, clip=clip, scale_layout="plain", return_dequant=True)
    w = ops.reorder_fp16_i4(xt, it, quant_mode=mode, clip=clip, scale_layout="plain", return_dequant=True, wide_codes=True)
    assert w[1].shape == (M, H - 128) and w[1].dtype == torch.int8
    assert np.array_equal(t2n(w[1]), wide_codes(ref["q4"]))
    for a, b in zip((p[0], p[2], p[3], p[4]), (w[0], w[2], w[3], w[4])):
        assert torch.equal(a, b)
    # the two fused producers share the store path
    wgt = torch.from_numpy((1 + 0.1 * np.random.default_rng(1).standard_normal(H)).astype(np.float16)).cuda()
    rp = ops.rmsnorm_fp16_i4(xt, wgt, it, 1e-5, quant_mode=mode, clip=clip, scale_layout="plain")
    rw = ops.rmsnorm_fp16_i4(xt, wgt, it, 1e-5, quant_mode=mode, clip=clip, scale_layout="plain", wide_codes=True)
    assert np.array_equal(t2n(rw[1]), wide_codes(O.unpack_int4(t2n(rp[1]).view(np.uint8))))
    b = torch.from_numpy(rand_act(M, H, seed=7, outliers=False)).cuda()
    sp = ops.activate_fp16_i4(xt, b, quant_mode=mode, clip=clip, scale_layout="plain")
    sw = ops.activate_fp16_i4(xt, b, quant_mode=mode, clip=clip, scale_layout="plain", wide_codes=True)
    assert np.array_equal(t2n(sw[1]), wide_codes(O.unpack_int4(t2n(sp[1]).view(np.uint8))))


@pytest.mark.parametrize("M,H", [(1, 256), (9, 1024), (33, 4096), (5, 5120), (2, 16384), (600, 512)])
@pytest.mark.parametrize("mode,clip", [("sim", 0.9), ("kernel", 1.0)])
def test_add_rmsnorm_is_add_then_rmsnorm(M, H, mode, clip):
    """atom_add_rmsnorm_reorder_quant_f16 == one fp16 add per element (torch half add) followed by exactly
    atom_rmsnorm_reorder_quant_f16 (itself bit-exact against the oracle above); also in place."""
    ops = _ops()
    g = np.random.default_rng(M + H)
    x = torch.from_numpy(rand_act(M, H, seed=M + H)).cuda()
    res = torch.from_numpy((g.standard_normal((M, H)) * 3).astype(np.float16)).cuda()
    w = torch.from_numpy((1 + 0.1 * g.standard_normal(H)).astype(np.float16)).cuda()
    idx = torch.from_numpy(g.permutation(H).astype(np.int16)).cuda()
    s_ref = x + res
    want = ops.rmsnorm_fp16_i4(s_ref, w, idx, 1e-5, quant_mode=mode, clip=clip, scale_layout="plain", return_dequant=True)
    got = ops.add_rmsnorm_fp16_i4(x, res, w, idx, 1e-5, quant_mode=mode, clip=clip, scale_layout="plain", return_dequant=True)
    assert torch.equal(got[0], s_ref)
    for a, b in zip(got[1:], want):
        assert torch.equal(a, b)
    oracle = O.rmsnorm_reorder_quant(t2n(s_ref), t2n(w), 1e-5, t2n(idx), mode, clip)
    assert np.array_equal(O.unpack_int4(t2n(got[2]).view(np.uint8)), oracle["q4"])
    res2 = res.clone()
    got2 = ops.add_rmsnorm_fp16_i4(x, res2, w, idx, 1e-5, inplace=True, quant_mode=mode, clip=clip, scale_layout="plain")
    assert got2[0] is res2 and torch.equal(res2, s_ref) and torch.equal(got2[2], want[1])


def test_f6_codes_from_all_three_quantisers():
    """ATOM_QUANT_F6_CODES: the BF6 group-major buffer holds the same codes as the packed call (decoded field by field;
    -0 and +0 are the same code value) and the scale of every (row, group) at byte 96 of its row."""
    from tests.helpers import f6_codes, f6_fields
    ops = _ops()
    M, H = 300, 1024
    x = torch.from_numpy(rand_act(M, H, seed=5)).cuda()
    b = torch.from_numpy(rand_act(M, H, seed=6, outliers=False)).cuda()
    w = torch.from_numpy((1 + 0.1 * np.random.default_rng(2).standard_normal(H)).astype(np.float16)).cuda()
    idx = torch.from_numpy(np.random.default_rng(3).permutation(H).astype(np.int16)).cuda()
    calls = [lambda **k: ops.reorder_fp16_i4(x, idx, scale_layout="plain", **k),
             lambda **k: ops.rmsnorm_fp16_i4(x, w, idx, 1e-5, scale_layout="plain", **k),
             lambda **k: ops.activate_fp16_i4(x, b, scale_layout="plain", **k),
             lambda **k: ops.add_rmsnorm_fp16_i4(x, b, w, idx, 1e-5, scale_layout="plain", **k)[1:]]
    for mode, clip in (("sim", 0.9), ("kernel", 1.0)):
        for call in calls:
            p = call(quant_mode=mode, clip=clip)
            f = call(quant_mode=mode, clip=clip, wide_codes="f6")
            assert f[1].shape == (H // 128 - 1, 512, 104) and f[1].dtype == torch.uint8
            want = f6_codes(O.unpack_int4(t2n(p[1]).view(np.uint8)), t2n(p[3]).T)
            got = t2n(f[1])
            assert np.array_equal(f6_fields(got[:, :M]), f6_fields(want[:, :M]))
            assert np.array_equal(got[:, :M, 96:], want[:, :M, 96:])
            for a, c in zip((p[0], p[2], p[3]), (f[0], f[2], f[3])):
                assert torch.equal(a, c)
