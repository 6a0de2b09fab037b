"""gate_proj + up_proj + SiLU x up + group quantiser as ONE launch (atom_gemm_w4a4_silu_mul_quant_f6), against the
three launches it replaces -- two W4A4 GEMMs with fp16 outputs and atom_silu_mul_quant_f16 (punica/models/llama.py:85-87;
model/qLlamaLayer.py:345-351) -- bit for bit: keeper codes, BF6 code fields, in-row and array scales, the fake-quant tensor."""
import numpy as np
import pytest
import torch

from tests.helpers import f6_fields, t2n

pytestmark = pytest.mark.gpu


def _weights(ops, N, K, g, scale):
    W = (torch.randn((N, K), device="cuda", generator=g) * scale).half()
    return ops.quant_weight_w4(W, 0.85, 2)


@pytest.mark.parametrize("M,N,K,order", [(512, 256, 384, 1), (1000, 1408, 640, 1), (2048, 11008, 4096, 1), (512, 11008, 4096, 1),
                                         (300, 512, 1152, 1), (512, 1024, 4096, 1), (1100, 2048, 1152, 2)])
@pytest.mark.parametrize("mode,clip", [("sim", 0.9), ("kernel", 1.0)])
@pytest.mark.parametrize("layout", ["plain", "ref"])
def test_fused_gate_up_equals_three_launches(M, N, K, order, mode, clip, layout):
    """order = the summation order of the stand-alone gate / up GEMMs (atom_gemm_w4a4_f6_order).  The fused launch always sums the K
    steps in order: where the stand-alone GEMMs do too (order 1) everything is bit-identical; where they add two / four ordered K
    ranges (few-tile shapes, which the module wrappers do not fuse) a gate / up value can differ in its last fp16 bit, so the bound is
    a stated fraction of codes off by one step."""
    from atom_amd import ops
    if layout == "ref" and N > 2000:
        pytest.skip("layout coverage on the small shapes")
    assert ops.L.lib().atom_gemm_w4a4_f6_order(M, N, K) == order
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = (torch.randn((M, K), device="cuda", generator=g)).half()
    x[:, -128:] *= 12
    gate, up = _weights(ops, N, K, g, 0.03), _weights(ops, N, K, g, 0.03)
    a = ops.reorder_fp16_i4(x, None, quant_mode="sim", clip=0.9, scale_layout=layout, wide_codes="f6")   # (o8, o6, s8, s4)
    # the three launches (F6 operands, fp16 outputs)
    def gemm(w):
        return ops.dense_layer_gemm_i4_fp16(a[1], ops.repack_weight_f6(w[0], w[2]), a[3], w[2], a[0], w[1], a[2], w[3],
                                            scale_layout=layout, a_wide="f6")
    y_g, y_u = gemm(gate), gemm(up)
    want = ops.activate_fp16_i4(y_g, y_u, quant_mode=mode, clip=clip, scale_layout=layout, return_dequant=True, wide_codes="f6")
    fused = ops.fuse_gate_up_weights(gate, up)
    got = ops.gate_up_silu_quant_f6(a[1], a[0], a[2], fused, quant_mode=mode, clip=clip, scale_layout=layout, return_dequant=True)
    if order != 1:
        g6, w6 = f6_fields(t2n(got[1])[:, :M]).astype(np.int32), f6_fields(t2n(want[1])[:, :M]).astype(np.int32)
        assert (g6 != w6).mean() <= 2e-3 and (got[0] != want[0]).float().mean().item() <= 2e-2
        d = (got[4].float() - want[4].float()).abs().max().item()
        from oracle import atom_oracle as O
        s4 = t2n(want[3]) if layout == "plain" else O.scales_from_ref_layout(t2n(want[3]), M)   # (the replicated layout has unwritten gaps)
        s8 = t2n(want[2]) if layout == "plain" else O.scales_from_ref_layout(t2n(want[2]), M)
        step = max(float(s4[..., :M].astype(np.float32).max()), float(s8[..., :M].astype(np.float32).max()))
        assert d <= 2.02 * step, (d, step)                                            # at most ~ one quantisation step (+ a scale ulp)
        return
    assert torch.equal(got[0], want[0]), "keeper INT8 codes"
    g6, w6 = t2n(got[1])[:, :M], t2n(want[1])[:, :M]
    assert np.array_equal(f6_fields(g6), f6_fields(w6)), "BF6 code fields"
    assert np.array_equal(g6[:, :, 96:], w6[:, :, 96:]), "in-row scales (fp16 + fp32)"
    if layout == "plain":
        assert torch.equal(got[2][:M], want[2][:M]) and torch.equal(got[3][:, :M], want[3][:, :M])
    else:
        from oracle import atom_oracle as O
        assert np.array_equal(O.scales_from_ref_layout(t2n(got[2]), M).view(np.uint16), O.scales_from_ref_layout(t2n(want[2]), M).view(np.uint16))
        assert np.array_equal(O.scales_from_ref_layout(t2n(got[3]), M).view(np.uint16), O.scales_from_ref_layout(t2n(want[3]), M).view(np.uint16))
    assert torch.equal(got[4].view(torch.int16), want[4].view(torch.int16)), "fake-quant fp16 tensor"
    # and the operand is usable as is: down_proj on the fused output == down_proj on the three-launch output
    down = _weights(ops, 256, N, g, 0.03)
    b6 = ops.repack_weight_f6(down[0], down[2])
    z1 = ops.dense_layer_gemm_i4_fp16(got[1], b6, got[3], down[2], got[0], down[1], got[2], down[3], scale_layout=layout, a_wide="f6")
    z2 = ops.dense_layer_gemm_i4_fp16(want[1], b6, want[3], down[2], want[0], down[1], want[2], down[3], scale_layout=layout, a_wide="f6")
    assert torch.equal(z1, z2)


def test_fused_gate_up_argument_checks():
    from atom_amd import ops
    from atom_amd._lib import AtomHipError
    g = torch.Generator(device="cuda").manual_seed(1)
    gate, up = _weights(ops, 256, 384, g, 0.03), _weights(ops, 256, 384, g, 0.03)
    fused = ops.fuse_gate_up_weights(gate, up)
    x = torch.randn((300, 384), device="cuda", generator=g).half()
    a = ops.reorder_fp16_i4(x, None, quant_mode="sim", clip=0.9, scale_layout="plain", wide_codes="f6")
    with pytest.raises(AtomHipError):
        ops.gate_up_silu_quant_f6(a[1], a[0], a[2], fused, quant_mode="sim", clip=1.5, scale_layout="plain")
    bad = dict(fused, n_inter=192)
    with pytest.raises((AtomHipError, AssertionError)):
        ops.gate_up_silu_quant_f6(a[1], a[0], a[2], bad, quant_mode="sim", clip=0.9, scale_layout="plain")


def test_e2e_llama_mlp_fused_path_equals_reference_call_order():
    """atom_amd.e2e.LlamaMLP (the mirror of punica's LlamaMLP, llama.py:71-87): from 512 tokens it runs gate/up/activate as one
    launch; the result must equal the reference's call order -- down(activate_fp16_i4(gate(x), up(x))) -- bit for bit."""
    import types
    from atom_amd.e2e import LlamaMLP, LlamaRMSNormInt4
    torch.manual_seed(3)
    H, I, M = 2048, 2816, 2048     # (sizes at which every unfused GEMM runs a tile kernel without split-K: same summation order)
    cfg = types.SimpleNamespace(hidden_size=H, intermediate_size=I)
    norm = LlamaRMSNormInt4(H, eps=1e-5).cuda()
    mlp = LlamaMLP(cfg).cuda()
    mlp.gate_proj.load_fp16_weight((torch.randn(I, H) * 0.05).half().cuda())
    mlp.up_proj.load_fp16_weight((torch.randn(I, H) * 0.05).half().cuda())
    mlp.down_proj.load_fp16_weight((torch.randn(H, I) * 0.03).half().cuda())
    a = norm((torch.randn(M, H) * 2).half().cuda())
    fused = mlp(a)
    mlp.FUSED_MIN_ROWS = 1 << 30
    plain = mlp(a)
    assert torch.equal(fused, plain)


def test_qllama_mlp_fused_path_equals_unfused():
    """atom_amd.model.QLlamaMLP (drop-in for model/qLlamaLayer.py:306-352) with F6 activation codes: fused launch == gate_proj,
    up_proj, act_fn x up, quantiser as separate modules, bit for bit (output and the codes handed to down_proj)."""
    from functools import partial
    import types
    from atom_amd.model import quant, qLlamaLayer
    args = types.SimpleNamespace(wbits=4, abits=4, a_sym=True, w_sym=True, act_group_size=128, weight_group_size=128,
                                 weight_channel_group=2, keeper=128, keeper_precision=3, a_clip_ratio=0.9, w_clip_ratio=0.85,
                                 kv_clip_ratio=1.0, tiling=0, exponential=False, quant_type="int", static=False, reorder=True,
                                 kv_cache=True)
    torch.manual_seed(5)
    H, I, M = 512, 1408, 640
    lin = lambda i, o, s: torch.nn.Linear(i, o, bias=False).half()
    orig = types.SimpleNamespace(gate_proj=lin(H, I, 0), up_proj=lin(H, I, 0), down_proj=lin(I, H, 0), act_fn=torch.nn.SiLU())
    for l, sc in ((orig.gate_proj, 0.05), (orig.up_proj, 0.05), (orig.down_proj, 0.03)):
        l.weight.data = (torch.randn_like(l.weight.float()) * sc).half()
    m = qLlamaLayer.QLlamaMLP(orig, args).to("cuda")
    for l in (m.gate_proj, m.up_proj, m.down_proj):
        l.quant()
    m.act_quant.configure(partial(quant.quantize_activation_wrapper, args=args), None)
    x = quant.hip_act_quant((torch.randn(2, M // 2, H, device="cuda") * 1.5).half(), args)
    assert quant.get_codes(x).wide == "f6"
    seen = {}
    def pre(mod, inp):
        seen[len(seen)] = (inp[0].clone(), quant.get_codes(inp[0]))
    h = m.down_proj.register_forward_pre_hook(pre)
    y1 = m(x)
    m.FUSED_MIN_ROWS = 1 << 30
    y2 = m(x)
    h.remove()
    assert torch.equal(y1, y2)
    (t1, c1), (t2, c2) = seen[0], seen[1]
    assert torch.equal(t1.view(torch.int16), t2.view(torch.int16)) and torch.equal(c1.o8, c2.o8) and torch.equal(c1.s4[:, :M], c2.s4[:, :M])
    assert np.array_equal(f6_fields(t2n(c1.o4)[:, :M]), f6_fields(t2n(c2.o4)[:, :M]))
