"""Pin oracle/atom_oracle.py against golden vectors produced by the unmodified reference
(tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import atom_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _bits(a):
    return np.asarray(a, dtype=np.float16).view(np.uint16)


@pytest.mark.parametrize("bits,clip", [(4, 0.9), (4, 0.85), (4, 1.0), (8, 1.0)])
def test_quantize_tensor_kat(golden_dir, bits, clip):
    """quant.py:118-183 incl. all-zero groups, the 1e-5 clamp, fp16 max, exact .5 ties."""
    z = _load(golden_dir, "kat_quantize_tensor.npz")
    q, s = O.quant_groups_sim(z["v"], bits, clip)
    got = O.dequant_sim(q, s)
    assert np.array_equal(_bits(got), _bits(z[f"out_b{bits}_c{clip}"]))
    assert q.min() >= -(2 ** (bits - 1)) and q.max() <= 2 ** (bits - 1) - 1


def test_weight_quant_c1_and_odd(golden_dir):
    """QLinearLayer.quant (qLinearLayer.py:42-78): integer-domain restatement reproduces Wq bit-exactly."""
    for name in ["c1_qlinear_16x512x512.npz", "weight_quant_256x640.npz"]:
        z = _load(golden_dir, name)
        w = O.quant_weight_sim(z["W"], 0.85, 2)
        assert np.array_equal(_bits(w["wq"]), _bits(z["Wq"])), name
        # pairs of output channels share a scale (quant.py:86-87)
        assert np.array_equal(w["s4"][:, 0::2], w["s4"][:, 1::2])


def test_act_quant_c1(golden_dir):
    z = _load(golden_dir, "c1_qlinear_16x512x512.npz")
    t = O._quant_row_tail(z["x"], "sim", 0.9)
    assert np.array_equal(_bits(O.act_dequant_sim(t)), _bits(z["xq"]))


def test_reorder_quant(golden_dir):
    z = _load(golden_dir, "reorder_quant_9x1024.npz")
    t = O.reorder_quant(z["x"], z["idx"], "sim", 0.9)
    assert np.array_equal(_bits(O.act_dequant_sim(t)), _bits(z["xq"]))


def test_rmsnorm_quant(golden_dir):
    z = _load(golden_dir, "rmsnorm_quant_7x1024.npz")
    y = O.rmsnorm_f16(z["x"], z["w"], float(z["eps"]), "sim")
    # torch.rsqrt / mean may differ from our 1/sqrt + f64 sum in the last fp32 bit -> allow 1 fp16 ulp
    # on a tiny fraction of the normalised values; everything downstream of equal values is bit-exact.
    d = np.abs(_bits(y).astype(np.int32) - _bits(z["normed"]).astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3
    t = O._quant_row_tail(z["normed"][:, z["idx"].astype(np.int64)], "sim", 0.9)
    assert np.array_equal(_bits(O.act_dequant_sim(t)), _bits(z["xq"]))


def test_silu_mul_quant(golden_dir):
    z = _load(golden_dir, "silu_mul_quant_5x1408.npz")
    p = O.silu_mul(z["a"], z["b"], "sim")
    d = np.abs(_bits(p).astype(np.int32) - _bits(z["prod"]).astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3          # expf implementations differ by ulps
    t = O._quant_row_tail(z["prod"], "sim", 0.9)
    assert np.array_equal(_bits(O.act_dequant_sim(t)), _bits(z["xq"]))


def test_gemm_matches_simulated_path_c1(golden_dir):
    """North-star tolerance: integer-domain GEMM vs the reference's F.linear on fake-quant operands
    (qLinearLayer.py:32-35) within 1e-2 relative."""
    z = _load(golden_dir, "c1_qlinear_16x512x512.npz")
    a = O._quant_row_tail(z["x"], "sim", 0.9)
    w = O.quant_weight_sim(z["W"], 0.85, 2)
    args = (a["q4"], w["q4"], a["s4"], w["s4"], a["q8"], w["q8"], a["s8"], w["s8"])
    d_exact = O.gemm_w4a4_exact(*args)
    d_ref = O.gemm_w4a4_ref(*args).astype(np.float64)
    y = z["y"].astype(np.float64)
    scale = np.sqrt((y ** 2).mean())
    assert np.abs(d_exact - y).max() <= 1e-2 * scale
    assert np.abs(d_ref - y).max() <= 1e-2 * scale
    assert np.linalg.norm(d_exact - y) / np.linalg.norm(y) < 1e-3
    # and our own numpy sim_linear equals torch's F.linear up to fp16 rounding of an fp32 sum
    y2 = O.sim_linear(z["xq"], z["Wq"]).astype(np.float64)
    assert np.abs(y2 - y).max() <= 2e-3 * scale


def test_scale_layout_roundtrip():
    """scale_index / SCALE_SIZE_A (Reorder.cuh:39-50, ops/__init__.py:137-138)."""
    assert O.scale_size(1) == 8 and O.scale_size(7) == 56 and O.scale_size(16) == 64
    assert O.scale_size(4096) == 4 * 4096
    for M in [1, 7, 8, 9, 15, 16, 17, 33, 100]:
        idx = [O.scale_index(r) + 2 * j for r in range(M) for j in range(4)]
        assert len(set(idx)) == 4 * M and max(idx) < O.scale_size(M)
        s = (np.arange(3 * M).reshape(3, M) + 1).astype(np.float16)
        assert np.array_equal(O.scales_from_ref_layout(O.scales_to_ref_layout(s), M), s)


def test_pack_roundtrip():
    c = np.arange(-8, 8, dtype=np.int8).reshape(1, 16)
    p = O.pack_int4(c)
    assert p[0, 0] == ((-8) & 0xF) | (((-7) & 0xF) << 4)
    assert np.array_equal(O.unpack_int4(p), c)


def test_kernel_mode_matches_reference_cpu_golden_semantics():
    """The kernel-flavoured tail (x * (1/s), half-away) vs the reference tests' CPU golden formula
    (x / s, std::round) -- test_Reorder.cu:41-112 -- within the reference's own tolerance:
    ints +-1, scales 1e-3 (test_Reorder.cu:261-328)."""
    rng = np.random.default_rng(0)
    x = ((rng.integers(0, 10, size=(64, 4096)) / 10.0)).astype(np.float16)   # (rand()%10)/10
    idx = rng.permutation(4096).astype(np.int16)
    t = O.reorder_quant(x, idx, "kernel", 1.0)
    y = x[:, idx.astype(np.int64)].astype(np.float32)
    body = y[:, :-128].reshape(64, 31, 128)
    s = np.abs(body).max(-1) / np.float32(7)
    q = np.clip(np.sign(body) * np.floor(np.abs(body / s[..., None]) + 0.5), -8, 7).reshape(64, -1)
    assert np.abs(q - t["q4"]).max() <= 1
    assert np.abs(s - t["s4"].astype(np.float32)).max() <= 1e-3


def _dequant_weight(r):
    G = r["s4"].shape[0]
    N = r["q4"].shape[0]
    w = np.empty((N, G * 128 + 128), dtype=np.float16)
    for g in range(G):
        w[:, g * 128:(g + 1) * 128] = O.dequant_sim(r["q4"][:, g * 128:(g + 1) * 128], r["s4"][g])
    w[:, G * 128:] = O.dequant_sim(r["q8"], r["s8"])
    return w


def test_pack_recovers_reference_gptq_codes(golden_dir):
    """The packer's CPU restatement against what the UNMODIFIED reference GPTQ wrote (gptq.py:197-331, fixture made by
    tests/golden/gen_golden_gptq.py): every block is on the grid, the recovered codes are the reference's
    round(q/scale_fp32) (gptq.py:38-39) and codes*scale_fp16 reproduces the stored weight to 2^-9."""
    z = _load(golden_dir, "gptq_layer_64x512.npz")
    q, s32, cg = z["q"], z["s32"], int(z["channel_group"])
    r = O.pack_weight_fq(q, cg)
    assert r["bad"] == 0
    for g in range(s32.shape[0]):
        blk = slice(g * 128, (g + 1) * 128)
        assert np.array_equal(r["q4"][:, blk], O.gptq_codes(q[:, blk], s32[g], 4, cg))
        assert np.abs(r["s4"][g].astype(np.float32) / np.repeat(s32[g], cg) - 1).max() < 2.0 ** -10
    assert np.array_equal(r["s4"][:, 0::2], r["s4"][:, 1::2])
    assert np.abs(r["q8"]).max(axis=1).min() == 127            # INT8 keeper, per-row symmetric (gptq.py:316-325)
    w = _dequant_weight(r).astype(np.float32)
    assert np.all(np.abs(w - q.astype(np.float32)) <= np.abs(q.astype(np.float32)) * 2.0 ** -9 + 2.0 ** -24)


def test_pack_is_exact_on_rtn_weights_and_flags_off_grid(golden_dir):
    z = _load(golden_dir, "weight_quant_256x640.npz")
    ref = O.quant_weight_sim(z["W"], 0.85, 2)
    r = O.pack_weight_fq(z["Wq"], 2)
    assert r["bad"] == 0
    assert np.array_equal(_bits(_dequant_weight(r)), _bits(z["Wq"]))     # the reference's own QLinearLayer.quant output
    same = np.array_equal(r["q4"], ref["q4"]) and np.array_equal(_bits(r["s4"]), _bits(ref["s4"]))
    assert same or np.array_equal(_bits(_dequant_weight(r)), _bits(ref["wq"]))
    n, k = z["W"].shape
    assert O.pack_weight_fq(z["W"], 2)["bad"] == (n // 2) * ((k - 128) // 128) + n   # unquantised: every block off grid


def test_kv_oracle_self_consistency():
    """The INT4 paged-KV restatement (pinned against the reference's CPU implementations in tests/test_oracle_ref.py) against an independent dense
    evaluation: append through the page tables, then decode == plain softmax attention over the de-quantised, RoPE'd
    rows gathered straight from the inputs."""
    g = np.random.default_rng(0)
    C, L, N, P, D = 12, 2, 4, 16, 128
    data = np.zeros((C, L, 2, N, P, D // 2), np.uint8)
    param = np.zeros((C, L, 2, N, P, 2), np.float16)
    seqs = [37, 5, 16]
    perm = list(g.permutation(C))
    indptr, indices, lpo = [0], [], []
    for s in seqs:
        nb = -(-s // P)
        indices += [perm.pop() for _ in range(nb)]
        indptr.append(indptr[-1] + nb)
        lpo.append((s - 1) % P + 1)
    T = sum(seqs)
    ai = np.cumsum([0] + seqs)
    k = g.integers(0, 256, (T, N, D // 2), dtype=np.uint8)
    v = g.integers(0, 256, (T, N, D // 2), dtype=np.uint8)
    kp = (g.random((T, N, 2)) * 0.1 + 0.01).astype(np.float16)
    vp = (g.random((T, N, 2)) * 0.1 + 0.01).astype(np.float16)
    O.kv_append_i4(data, param, indptr, indices, lpo, k, v, kp, vp, 1, ai)
    assert not data[:, 0].any()                                   # other layer untouched
    q = g.standard_normal((3, N, D)).astype(np.float16)
    o = O.batch_decode_i4(q, data, param, indptr, indices, lpo, 1)
    for b, s in enumerate(seqs):
        for h in range(N):
            sl = slice(ai[b], ai[b + 1])
            kf = O._rope_llama(O._dequant_u4_rows(k[sl, h], kp[sl, h]), np.arange(s))
            vf = O._dequant_u4_rows(v[sl, h], vp[sl, h])
            qf = O._rope_llama(q[b, h].astype(np.float64)[None], [s - 1])[0]
            w = np.exp(kf @ qf / np.sqrt(D))
            assert np.allclose(o[b, h], (w / w.sum()) @ vf, rtol=1e-10, atol=1e-12)
    # RoPE is a rotation of the (i, i+64) pairs: norms are preserved, position 0 is the identity
    x = g.standard_normal((5, D))
    assert np.allclose(np.linalg.norm(O._rope_llama(x, np.arange(5)), axis=1), np.linalg.norm(x, axis=1))
    assert np.allclose(O._rope_llama(x[:1], [0]), x[:1])


def test_kv_fake_quant_matches_reference(golden_dir):
    """quantize_attn_k/v_wrapper (model/quant.py:233-257): the FP16-opmath restatement reproduces the unmodified reference
    bit for bit (tests/golden/gen_golden_kv.py), clip and 8-bit variants included."""
    z = _load(golden_dir, "kv_fake_quant.npz")
    for key in z:
        if key == "x":
            continue
        bits, clip = int(key.split("_b")[1].split("_")[0]), float(key.split("_c")[1])
        assert np.array_equal(_bits(O.kv_fake_quant_sim(z["x"], bits, clip)), _bits(z[key])), key
