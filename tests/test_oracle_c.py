"""The plain-C restatement (oracle/atom_oracle.c) against the numpy oracle (itself pinned to the reference goldens)."""
import numpy as np
import pytest

from oracle import atom_oracle as O
from tests import c_oracle as C
from tests.helpers import bits16, rand_act, rand_gemm_operands


def test_half_conversions_exhaustive():
    L = C.lib()
    allh = np.arange(65536, dtype=np.uint16)
    want = allh.view(np.float16).astype(np.float32)
    got = np.array([L.oracle_h2f(int(h)) for h in allh], dtype=np.float32)
    ok = (got == want) | (np.isnan(got) & np.isnan(want))
    assert ok.all()
    rng = np.random.default_rng(0)
    f = np.concatenate([rng.standard_normal(20000) * s for s in (1e-8, 1e-5, 1e-3, 1, 100, 7e4)]).astype(np.float32)
    f = np.concatenate([f, want[~np.isnan(want)], np.float32([65519.99, 65520, 6e-8, 2.98e-8, 2.99e-8, 0.0, -0.0])])
    with np.errstate(over="ignore"):
        w = f.astype(np.float16).view(np.uint16)
    g = np.array([L.oracle_f2h(float(v)) for v in f], dtype=np.uint16)
    assert np.array_equal(g, w)


def test_layout():
    L = C.lib()
    for r in range(300):
        assert L.oracle_scale_index(r) == O.scale_index(r)
        assert L.oracle_scale_size(r) == O.scale_size(r)


@pytest.mark.parametrize("sim,clip", [(1, 0.9), (0, 1.0), (0, 0.9), (1, 1.0)])
def test_reorder_and_rmsnorm_exact(sim, clip):
    mode = "sim" if sim else "kernel"
    x = rand_act(11, 1024, 3)
    x[2] = 0
    idx = np.random.default_rng(1).permutation(1024).astype(np.int16)
    w = (1 + 0.1 * np.random.default_rng(2).standard_normal(1024)).astype(np.float16)
    for op, b, ref in [(0, None, O.reorder_quant(x, idx, mode, clip)),
                       (1, w, O.rmsnorm_reorder_quant(x, w, 1e-5, idx, mode, clip))]:
        got = C.act_quant(op, x, b, idx, sim, clip, 1e-5)
        assert np.array_equal(got["q4"], ref["q4"]) and np.array_equal(got["q8"], ref["q8"])
        assert np.array_equal(bits16(got["s4"]), bits16(ref["s4"]))
        assert np.array_equal(bits16(got["s8"]), bits16(ref["s8"]))


@pytest.mark.parametrize("sim,clip", [(1, 0.9), (0, 1.0)])
def test_silu_mul_close(sim, clip):
    g = np.random.default_rng(5)
    a = (g.standard_normal((6, 1408)) * 2).astype(np.float16)
    b = (g.standard_normal((6, 1408)) * 2).astype(np.float16)
    ref = O.silu_mul_quant(a, b, "sim" if sim else "kernel", clip)
    got = C.act_quant(2, a, b, None, sim, clip)
    d = np.abs(got["q4"].astype(int) - ref["q4"])
    assert d.max() <= 1 and (d > 0).mean() < 2e-3            # glibc expf vs numpy exp: ulp-level differences


def test_gemm_contract_vs_exact():
    d = rand_gemm_operands(24, 64, 640, seed=9)
    D = C.gemm(O.pack_int4(d["qa4"]), O.pack_int4(d["qb4"]), d["sA"].T, d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"])
    ex = O.gemm_w4a4_exact(d["qa4"], d["qb4"], d["sA"], d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"])
    with np.errstate(over="ignore"):
        exh = ex.astype(np.float16)
    ulp = np.abs(bits16(D).astype(np.int32) - bits16(exh).astype(np.int32))
    assert ulp.max() <= 1 and (ulp > 0).mean() < 0.02        # FP32 accumulation vs FP64: at most the last fp16 bit


@pytest.mark.parametrize("K", [256, 640, 2176])
def test_gemm_decode_summation_order(K):
    """oracle_gemm_w4a4_f16_split (the decode-batch kernel's order: items dealt to 8 waves, partial sums added in wave
    order): identical to the contract order when one wave gets every item, within 1 fp16 ulp of exact otherwise."""
    d = rand_gemm_operands(9, 64, K, seed=K)
    args = (O.pack_int4(d["qa4"]), O.pack_int4(d["qb4"]), d["sA"].T, d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"])
    base = C.gemm(*args)
    assert np.array_equal(bits16(C.gemm(*args, nsplit=1)), bits16(base))
    split = C.gemm(*args, nsplit=8)
    ex = O.gemm_w4a4_exact(d["qa4"], d["qb4"], d["sA"], d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"])
    with np.errstate(over="ignore"):
        exh = ex.astype(np.float16)
    ulp = np.abs(bits16(split).astype(np.int32) - bits16(exh).astype(np.int32))
    assert ulp.max() <= 1
    G = (K - 128) // 128
    huge = C.gemm(*args, nsplit=2 * (G + 1))       # per = 1 -> every item its own wave: still a valid order
    assert np.abs(bits16(huge).astype(np.int32) - bits16(exh).astype(np.int32)).max() <= 1


@pytest.mark.parametrize("K,kgroups", [(640, 1), (1152, 2), (2176, 2), (2176, 4), (4096, 4), (1408, 4)])
def test_gemm_k_group_orders_c_vs_numpy(K, kgroups):
    """The summation orders of the tile kernels (one ordered sum; two / four ordered ranges of the K steps, atom_gemm_w4a4_f6_order)
    restated twice, independently: oracle/atom_oracle.c (gemm_core, nsplit = -kgroups) and numpy (gemm_w4a4_contract).  Bit for bit."""
    d = rand_gemm_operands(11, 128, K, seed=3 * K + kgroups)
    args = (O.pack_int4(d["qa4"]), O.pack_int4(d["qb4"]), d["sA"].T, d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"])
    c = C.gemm(*args, nsplit=-kgroups if kgroups > 1 else 1)
    n = O.gemm_w4a4_contract(d["qa4"], d["qb4"], d["sA"], d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"], kgroups=kgroups)
    assert np.array_equal(bits16(c), bits16(n))
    if kgroups > 1:                                            # a different order than the single sum, same value to 1 fp16 ulp
        base = C.gemm(*args)
        assert np.abs(bits16(c).astype(np.int32) - bits16(base).astype(np.int32)).max() <= 1
