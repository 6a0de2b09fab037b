"""GPU tests of atom_pack_weight_w4 -- the bridge that lets GPTQ-written weights (gptq.py:331) run on the W4A4 GEMM
(SURVEY 8(f) N2).  Checker: oracle.pack_weight_fq / gptq_codes and the reference-GPTQ fixture."""
import os
import types

import numpy as np
import pytest
import torch

from helpers import bits16, t2n
from oracle import atom_oracle as O

pytestmark = pytest.mark.gpu


def _ops():
    from atom_amd import ops
    return ops


def _dequant(b4, b8, sb, sb8):
    q4 = O.unpack_int4(t2n(b4)).astype(np.float32)
    s4 = t2n(sb).astype(np.float32)                       # [G, N]
    N, K4 = q4.shape
    w4 = (q4.reshape(N, -1, 128) * s4.T[:, :, None]).reshape(N, K4)
    w8 = t2n(b8).astype(np.float32) * t2n(sb8).astype(np.float32)[:, None]
    return np.concatenate([w4, w8], axis=1)


def test_pack_reference_gptq_fixture(golden_dir):
    ops = _ops()
    z = np.load(os.path.join(golden_dir, "gptq_layer_64x512.npz"))
    q, s32, cg = z["q"], z["s32"], int(z["channel_group"])
    b4, b8, sb, sb8, bad = ops.pack_weight_w4(torch.from_numpy(q).cuda(), cg)
    assert bad == 0
    codes = O.unpack_int4(t2n(b4))
    for g in range(s32.shape[0]):
        blk = slice(g * 128, (g + 1) * 128)
        assert np.array_equal(codes[:, blk], O.gptq_codes(q[:, blk], s32[g], 4, cg))
        assert np.abs(t2n(sb)[g].astype(np.float32) / np.repeat(s32[g], cg) - 1).max() < 2.0 ** -10
    ref = O.pack_weight_fq(q, cg)
    assert np.array_equal(codes, ref["q4"]) and np.array_equal(t2n(b8), ref["q8"])
    # scales: the kernel and the numpy restatement sum the least-squares terms in different orders -> <= 1 fp16 ulp apart
    assert np.abs(bits16(t2n(sb)).astype(np.int32) - bits16(ref["s4"]).astype(np.int32)).max() <= 1
    assert np.abs(bits16(t2n(sb8)).astype(np.int32) - bits16(ref["s8"]).astype(np.int32)).max() <= 1
    w = _dequant(b4, b8, sb, sb8)
    qf = q.astype(np.float32)
    assert np.all(np.abs(w - qf) <= np.abs(qf) * 2.0 ** -9 + 2.0 ** -24)


@pytest.mark.parametrize("N,K,cg", [(256, 640, 2), (512, 512, 2), (64, 4096, 1), (128, 11008, 2)])
def test_pack_roundtrip_of_quant_weight_is_exact(N, K, cg):
    """QLinearLayer.quant output (fp16 scales): pack(Wq) must reproduce Wq bit for bit, and feed the GEMM identically."""
    ops = _ops()
    W = torch.from_numpy((np.random.default_rng(N * 7 + K).standard_normal((N, K)) * 0.05).astype(np.float16)).cuda()
    b4, b8, sb, sb8, wq = ops.quant_weight_w4(W, 0.85, cg, return_fake_quant=True)
    p4, p8, ps, ps8, bad = ops.pack_weight_w4(wq, cg)
    assert bad == 0
    w = _dequant(p4, p8, ps, ps8).astype(np.float16)       # product exact in fp32 -> one rounding, as quant.py:181
    assert np.array_equal(bits16(w), bits16(t2n(wq)))
    x = torch.randn(24, K, device="cuda", dtype=torch.float16)
    o8, o4, s8, s4 = ops.reorder_fp16_i4(x, None, scale_layout="plain")[:4]
    d0 = ops.dense_layer_gemm_i4_fp16(o4, b4, s4, sb, o8, b8, s8, sb8, scale_layout="plain")
    d1 = ops.dense_layer_gemm_i4_fp16(o4, p4, s4, ps, o8, p8, s8, ps8, scale_layout="plain")
    assert torch.allclose(d0.float(), d1.float(), rtol=2e-3, atol=2e-3 * float(d0.float().abs().max()))


def test_pack_fp32_scale_weights_full_size():
    """GPTQ-style synthetic weight at the headline size: half(s32*c) with FP32 scales per (2 rows x 128 cols)."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(5)
    N, K = 4096, 4096
    K4, G = K - 128, (K - 128) // 128
    c4 = torch.randint(-8, 8, (N, K4), device="cuda", generator=g)
    s4 = (torch.rand((N // 2, G), device="cuda", generator=g) * 0.02 + 0.001).repeat_interleave(2, 0)   # [N, G] fp32
    c8 = torch.randint(-127, 128, (N, 128), device="cuda", generator=g)
    c8[:, 0] = 127
    s8 = torch.rand((N, 1), device="cuda", generator=g) * 0.002 + 0.0001
    w = torch.cat([(c4.float().view(N, G, 128) * s4[:, :, None]).view(N, K4), c8.float() * s8], 1).half()
    b4, b8, sb, sb8, bad = ops.pack_weight_w4(w, 2)
    assert bad == 0
    rec = torch.from_numpy(_dequant(b4, b8, sb, sb8)).cuda()
    wf = w.float()
    assert bool(torch.all((rec - wf).abs() <= wf.abs() * 2.0 ** -9 + 2.0 ** -24))
    assert np.array_equal(t2n(b8), t2n(c8).astype(np.int8))
    # code ambiguity only by a common integer factor of a block; with 256 uniform codes per block that never happens
    assert np.array_equal(O.unpack_int4(t2n(b4)), t2n(c4).astype(np.int8))


def test_pack_flags_off_grid_and_edges():
    ops = _ops()
    from atom_amd._lib import AtomHipError
    W = (torch.randn(64, 512, device="cuda") * 0.05).half()
    with pytest.raises(AtomHipError):
        ops.pack_weight_w4(W, 2)
    *_, bad = ops.pack_weight_w4(W, 2, strict=False)
    assert bad == 32 * 3 + 64
    z = torch.zeros(64, 512, device="cuda", dtype=torch.float16)          # all-zero blocks: codes 0, scale 0
    b4, b8, sb, sb8, bad = ops.pack_weight_w4(z, 2)
    assert bad == 0 and int(b4.max()) == 0 and int(b8.abs().max()) == 0 and float(sb.abs().max()) == 0.0
    with pytest.raises(AtomHipError):
        ops.pack_weight_w4(torch.zeros(63, 512, device="cuda", dtype=torch.float16), 2)     # odd N
    with pytest.raises(AtomHipError):
        ops.pack_weight_w4(torch.zeros(64, 500, device="cuda", dtype=torch.float16), 2)     # K % 128


def test_qlinear_gptq_flow_runs_on_the_hip_gemm(golden_dir):
    """modelutils_llama.py:224-258 in miniature: forward with the unquantised weight (Hessian collection) -> F.linear;
    GPTQ assigns ``layer.weight.data = Q`` (gptq.py:331) -> the next forward packs lazily and runs atom_gemm_w4a4_f16."""
    from atom_amd.model import quant as Q
    from atom_amd.model.qLinearLayer import QLinearLayer
    z = np.load(os.path.join(golden_dir, "gptq_layer_64x512.npz"))
    args = types.SimpleNamespace(wbits=4, abits=4, a_sym=True, w_sym=True, act_group_size=128, weight_group_size=128,
                                 weight_channel_group=2, keeper=128, keeper_precision=3, a_clip_ratio=0.9,
                                 w_clip_ratio=0.85, kv_clip_ratio=1.0, tiling=0, exponential=False, quant_type="int",
                                 static=False, reorder=True)
    lin = torch.nn.Linear(512, 64, bias=False).half()
    lin.weight.data = torch.from_numpy(z["w0"])
    layer = QLinearLayer(lin, args).to("cuda")
    x = torch.randn(40, 512, device="cuda", dtype=torch.float16)
    x[:, -128:] *= 10
    xq = Q.quantize_activation_wrapper(x.clone(), args)
    assert Q.get_codes(xq) is not None
    y0 = layer(xq)
    assert layer.packed_weight() is None                                   # unquantised weight: reference forward
    assert torch.equal(y0, torch.nn.functional.linear(xq, layer.weight))
    layer.weight.data = torch.from_numpy(z["q"]).cuda()                    # what GPTQ.fasterquant does
    y1 = layer(xq)
    assert layer.packed_weight() is not None
    ref = torch.nn.functional.linear(xq.float(), layer.weight.float()).detach()
    assert float((y1.float() - ref).abs().max()) <= 1e-2 * float(ref.abs().max())
    layer.cpu()
    layer.to("cuda")
    assert layer.packed_weight() is not None and torch.equal(layer(xq), y1)


def test_export_packed_loads_into_e2e_modules(tmp_path):
    """quantise with the B1 modules, save, load into the real-kernel module tree (same parameter names
    and shapes as the reference's LinearInt4 / LlamaRMSNormInt4) and get the same GEMM result."""
    from atom_amd import ops
    from atom_amd.e2e import LinearInt4, LlamaRMSNormInt4
    from atom_amd.model import export
    from atom_amd.model.qLinearLayer import QLinearLayer
    args = types.SimpleNamespace(wbits=4, abits=4, a_sym=True, w_sym=True, act_group_size=128, weight_group_size=128,
                                 weight_channel_group=2, keeper=128, keeper_precision=3, a_clip_ratio=0.9,
                                 w_clip_ratio=0.85, kv_clip_ratio=1.0, tiling=0, exponential=False, quant_type="int",
                                 static=False, reorder=True)
    torch.manual_seed(0)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.up_proj = QLinearLayer(torch.nn.Linear(512, 1408, bias=False).half(), args)
            self.down_proj = QLinearLayer(torch.nn.Linear(1408, 512, bias=False).half(), args)

    m = Tiny().cuda()
    for l in (m.up_proj, m.down_proj):
        l.to("cuda")
        l.quant()
    path = str(tmp_path / "tiny.safetensors")
    export.save_packed(m, path)
    sd = export.load_packed(path)
    assert set(sd) == {f"{n}.{k}" for n in ("up_proj", "down_proj") for k in ("weight_int4", "weight_int8", "scale_int4", "scale_int8")}

    class TinyE2E(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.up_proj = LinearInt4(512, 1408, out_dtype="fp16")
            self.down_proj = LinearInt4(1408, 512, out_dtype="fp16")

    e = TinyE2E()
    missing, unexpected = e.load_state_dict(sd, strict=True), None
    e = e.cuda()
    x = torch.randn(33, 512, device="cuda", dtype=torch.float16)
    act = ops.reorder_fp16_i4(x, None)
    y = e.up_proj(act)
    b4, b8, sb, sb8 = m.up_proj.packed_weight()
    want = ops.dense_layer_gemm_i4_fp16(act[1], b4, act[3], sb, act[0], b8, act[2], sb8)
    assert torch.equal(y, want)
