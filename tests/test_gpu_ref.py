"""The HIP activation-quant kernels in kernel mode (ATOM_QUANT_KERNEL, the arithmetic of the reference's CUDA kernels), through
the C ABI, against the reference's OWN CPU golden functions compiled from /root/reference (oracle/_ref/libatom_ref.so; see
tests/test_oracle_ref.py).  Bounds: the reference's own acceptance (codes +-1, test_Reorder.cu:269-318) tightened to a stated
flip fraction; scales within one fp16 ulp; the replicated scale layout and the nibble packing bit for bit."""
import numpy as np
import pytest
import torch

from oracle import atom_oracle as O
from tests import ref_golden as R
from tests.helpers import bits16, rand_act, t2n

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not R.available(), reason="oracle/_ref/libatom_ref.so not built (needs /root/reference)")]

SHAPES = [(1, 256), (37, 4096), (16, 11008), (8, 5120), (130, 4096)]


def _cmp(ref, outs, M, max_flip, scale_ulps):
    o8, o4, s8, s4 = [t2n(t) for t in outs[:4]]
    q4 = O.unpack_int4(o4.view(np.uint8))
    d4 = np.abs(ref["q4"].astype(np.int32) - q4)
    d8 = np.abs(ref["q8"].astype(np.int32) - o8)
    assert d4.max() <= 1 and d8.max() <= 1
    assert (d4 > 0).mean() <= max_flip and (d8 > 0).mean() <= max_flip * 20, ((d4 > 0).mean(), (d8 > 0).mean())
    # scales in the reference's replicated layout, compared where the reference writes them (all four replicas)
    rows = np.array([O.scale_index(r) for r in range(M)])
    G = ref["s4"].shape[1]
    ld = R.scale_size(M)
    for k in range(4):
        got4 = s4.reshape(G, -1)[:, rows + 2 * k]
        want4 = ref["raw_s4"].reshape(G, ld)[:, rows + 2 * k]
        assert np.abs(bits16(got4).astype(np.int32) - bits16(want4).astype(np.int32)).max() <= scale_ulps
        assert np.abs(bits16(s8[rows + 2 * k]).astype(np.int32) - bits16(ref["raw_s8"][rows + 2 * k]).astype(np.int32)).max() <= scale_ulps
    return (d4 > 0).mean()


@pytest.mark.parametrize("M,H", SHAPES)
def test_reorder_vs_reference_cpu_golden(M, H):
    from atom_amd import ops
    x = rand_act(M, H, seed=M + H)
    idx = np.random.default_rng(H).permutation(H).astype(np.int16)
    outs = ops.reorder_fp16_i4(torch.from_numpy(x).cuda(), torch.from_numpy(idx).cuda(), quant_mode="kernel", clip=1.0,
                               scale_layout="ref")
    _cmp(R.reorder(x, idx), outs, M, max_flip=5e-4, scale_ulps=0)


@pytest.mark.parametrize("M,H", SHAPES)
def test_activate_vs_reference_cpu_golden(M, H):
    from atom_amd import ops
    g = np.random.default_rng(M * 7 + H)
    a = (g.standard_normal((M, H)) * 2).astype(np.float16)
    b = g.standard_normal((M, H)).astype(np.float16)
    outs = ops.activate_fp16_i4(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), quant_mode="kernel", clip=1.0,
                                scale_layout="ref")
    _cmp(R.activate(a, b), outs, M, max_flip=2e-3, scale_ulps=1)


@pytest.mark.parametrize("M,H", SHAPES)
def test_rmsnorm_vs_reference_cpu_golden(M, H):
    from atom_amd import ops
    x = rand_act(M, H, seed=3 * M + H)
    g = np.random.default_rng(H + 1)
    w = (1 + 0.1 * g.standard_normal(H)).astype(np.float16)
    idx = g.permutation(H).astype(np.int16)
    outs = ops.rmsnorm_fp16_i4(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), torch.from_numpy(idx).cuda(), 1e-5,
                               quant_mode="kernel", clip=1.0, scale_layout="ref")
    _cmp(R.rmsnorm(x, w, 1e-5, idx), outs, M, max_flip=2e-3, scale_ulps=1)


@pytest.mark.parametrize("M,N,K,ws", [(40, 128, 640, False), (300, 384, 384, False), (40, 256, 640, True), (16, 4096, 1152, True),
                                      (256, 1024, 4096, False)])
def test_gemm_o4_ref_extrema_vs_reference_epilogue_code(M, N, K, ws):
    """SURVEY 8(a) a18, pinned: the library's ATOM_O4_REF_EXTREMA output equals, bit for bit, the reference's own epilogue code --
    DenseLayerGEMM_i4_o4.cu:722-786 applied around its host-callable local_max_min (:72-80), compiled from the reference source
    into oracle/_ref (oracle/ref/wrap.cpp: ref_cpu_o4_epilogue) -- on the GEMM's FP32 sums: the tile kernel (sums = the C contract,
    itself bit-exact vs the kernel) and the decode path (sums = atom_gemm_w4a4_f32).  Codes are compared wherever the reference's
    unclamped float -> int8 cast is defined (|quotient| < 127); (scale, zero) everywhere."""
    from atom_amd import ops
    from tests import c_oracle as C
    from tests.helpers import rand_gemm_operands, to_device
    d = rand_gemm_operands(M, N, K, seed=M + 7 * N + K)
    h = M // 2                                             # first half of the rows: non-negative sums (both epilogues agree there)
    d["qa4"][:h] = np.abs(d["qa4"][:h]); d["qa8"][:h] = np.abs(d["qa8"][:h].astype(np.int16)).clip(0, 127).astype(np.int8)
    d["qb4"] = np.abs(d["qb4"]); d["qb8"] = np.abs(d["qb8"].astype(np.int16)).clip(0, 127).astype(np.int8)
    dev = to_device(d, "plain")
    q, sz = ops.dense_layer_gemm_i4_o4(*dev, scale_layout="plain", use_workspace=ws, ref_extrema=True)
    if ws:
        assert ops.decode_gemm_fits(M, N, K)
        c32 = t2n(ops.dense_layer_gemm_i4_f32(*dev, scale_layout="plain"))
    else:
        c32 = C.gemm(O.pack_int4(d["qa4"]), O.pack_int4(d["qb4"]), d["sA"].T, d["sB"], d["qa8"], d["qb8"], d["sA8"], d["sB8"], fp32=True)
    want_q, want_sz = R.o4_epilogue(c32)
    assert np.array_equal(bits16(t2n(sz).reshape(M, N // 128, 2)), bits16(want_sz))
    ok = R.o4_in_range(c32)
    assert ok.mean() > 0.95 and (c32[h:] < 0).any()
    assert np.array_equal(t2n(q).reshape(M, -1, 64)[ok], want_q.reshape(M, -1, 64)[ok])
    # the default (intended min / max) epilogue is the reference code's on groups without negative sums
    q_def, sz_def = ops.dense_layer_gemm_i4_o4(*dev, scale_layout="plain", use_workspace=ws)
    pos = (c32.reshape(M, -1, 128) >= 0).all(-1) & ok
    assert pos.any()
    assert np.array_equal(t2n(q_def).reshape(M, -1, 64)[pos], want_q.reshape(M, -1, 64)[pos])
    assert np.array_equal(bits16(t2n(sz_def).reshape(M, N // 128, 2))[pos], bits16(want_sz)[pos])
