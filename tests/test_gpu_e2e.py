"""The reference's real-kernel call graph (punica LinearInt4 / LlamaRMSNormInt4 / LlamaMLP) on the HIP ops, checked op by
op against the oracle's kernel-flavoured restatement, plus C-ABI misuse."""
import types

import numpy as np
import pytest
import torch

from oracle import atom_oracle as O
from tests.helpers import t2n

pytestmark = pytest.mark.gpu


def test_rmsnorm_mlp_chain_kernel_mode():
    from atom_amd.e2e import LlamaMLP, LlamaRMSNormInt4
    torch.manual_seed(0)
    H, I, M = 512, 1408, 37
    cfg = types.SimpleNamespace(hidden_size=H, intermediate_size=I)
    norm = LlamaRMSNormInt4(H, eps=1e-5).cuda()
    norm.weight.data = (1 + 0.1 * torch.randn(H)).half().cuda()
    mlp = LlamaMLP(cfg).cuda()
    Wg, Wu = (torch.randn(I, H) * 0.05).half().cuda(), (torch.randn(I, H) * 0.05).half().cuda()
    Wd = (torch.randn(H, I) * 0.03).half().cuda()
    mlp.gate_proj.load_fp16_weight(Wg); mlp.up_proj.load_fp16_weight(Wu); mlp.down_proj.load_fp16_weight(Wd)
    x = (torch.randn(M, H) * 2).half().cuda()
    y = mlp(norm(x))
    assert y.shape == (M, H) and y.dtype == torch.float16

    # oracle, stage by stage (each stage fed with the HIP output of the previous one: teacher forcing)
    idx = t2n(norm.reorder_index)
    a = O.rmsnorm_reorder_quant(t2n(x), t2n(norm.weight), 1e-5, idx, "kernel", 1.0)
    o8, o4, s8, s4 = norm(x)
    assert np.array_equal(O.unpack_int4(t2n(o4).view(np.uint8)), a["q4"]) and np.array_equal(t2n(o8), a["q8"])
    wg, wu, wd = (O.quant_weight_sim(t2n(w), 0.85, 2) for w in (Wg, Wu, Wd))
    gemm = lambda act, w: O.gemm_w4a4_exact(act["q4"], w["q4"], act["s4"], w["s4"], act["q8"], w["q8"], act["s8"], w["s8"])
    g_hip, u_hip = mlp.gate_proj((o8, o4, s8, s4)), mlp.up_proj((o8, o4, s8, s4))
    for got, want in ((g_hip, gemm(a, wg)), (u_hip, gemm(a, wu))):
        assert np.abs(t2n(got).astype(np.float64) - want).max() <= 2e-3 * np.sqrt((want ** 2).mean()) + 1e-3 * np.abs(want).max()
    b = O.silu_mul_quant(t2n(g_hip), t2n(u_hip), "kernel", 1.0)
    act = __import__("atom_amd").ops.activate_fp16_i4(g_hip, u_hip)
    d4 = np.abs(O.unpack_int4(t2n(act[1]).view(np.uint8)).astype(int) - b["q4"])
    assert d4.max() <= 1 and (d4 > 0).mean() < 2e-3
    # final GEMM from the HIP activation codes
    bh = dict(q4=O.unpack_int4(t2n(act[1]).view(np.uint8)), q8=t2n(act[0]),
              s4=O.scales_from_ref_layout(t2n(act[3]), M).T, s8=O.scales_from_ref_layout(t2n(act[2]), M))
    want = gemm(bh, wd)
    assert np.abs(t2n(y).astype(np.float64) - want).max() <= 2e-3 * np.sqrt((want ** 2).mean()) + 1e-3 * np.abs(want).max()


def test_linear_int4_o4_output():
    from atom_amd.e2e import LinearInt4
    from atom_amd import ops
    torch.manual_seed(1)
    lin = LinearInt4(512, 256, out_dtype="int4").cuda().load_fp16_weight((torch.randn(256, 512) * 0.05).half().cuda())
    x = torch.randn(20, 512).half().cuda()
    q, sz = lin(ops.reorder_fp16_i4(x, torch.randperm(512).to(torch.int16).cuda()))
    assert q.shape == (20, 128) and q.dtype == torch.uint8 and sz.shape == (20, 4)


def test_c_abi_rejects_bad_arguments():
    """Validation happens before any launch and is reported as a status code, never as a fault."""
    import ctypes
    from atom_amd import _lib as L
    lib = L.lib()
    t = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    p = t.data_ptr()
    ok_args = [p, p, p, p, p, p, p, p, p]
    assert lib.atom_gemm_w4a4_f16(*ok_args, 4, 64, 256, 128, 128, 1, None) == 0
    assert lib.atom_gemm_w4a4_f16(None, *ok_args[1:], 4, 64, 256, 128, 128, 1, None) == -22        # null pointer
    assert lib.atom_gemm_w4a4_f16(*ok_args, 4, 64, 256, 64, 128, 1, None) == -33                   # group != 128
    assert lib.atom_gemm_w4a4_f16(*ok_args, 4, 64, 256, 128, 64, 1, None) == -33                   # keeper != 128
    assert lib.atom_gemm_w4a4_f16(*ok_args, 4, 64, 300, 128, 128, 1, None) == -33                  # K4 % 128 != 0
    assert lib.atom_gemm_w4a4_f16(*ok_args, 0, 64, 256, 128, 128, 1, None) == -33                  # M = 0
    assert lib.atom_gemm_w4a4_f16(*ok_args, 4, 64, 256, 128, 128, 7, None) == -22                  # bad layout enum
    assert lib.atom_gemm_w4a4_f16(p + 4, *ok_args[1:], 4, 64, 256, 128, 128, 1, None) == -14       # misaligned A4
    assert lib.atom_gemm_w4a4_o4(*ok_args, p, 4, 64, 256, 128, 128, 1, None) == -33                # o4 needs N % 128 == 0
    assert lib.atom_reorder_quant_f16(p, None, 4, 512, 5, 1.0, 1, p, p, p, p, None, None) == -22   # bad quant mode
    assert lib.atom_reorder_quant_f16(p, None, 4, 32768, 0, 1.0, 1, p, p, p, p, None, None) == -33 # hidden too large
    assert lib.atom_quant_weight_w4(p, 3, 256, 0.85, 2, p, p, p, p, None, None) == -33             # odd N
    assert lib.atom_quant_weight_w4(p, 4, 256, 0.85, 3, p, p, p, p, None, None) == -22             # channel_group 3
    # atom_gemm_w4a4_multi_q(q_op, x, x2, residual, residual_out, reorder_index, eps, clip, B4, sB, B8, sB8, out0..2, mask, add, M, N_seg, nseg, K, ...)
    q = p + (1 << 15)
    mq = lambda op, x2, res, res_out, M=1, idx=None: lib.atom_gemm_w4a4_multi_q(op, p, x2, res, res_out, idx, 1e-5, 1.0, p, p, p, p, q, None, None, 0, None,
                                                                                 M, 64, 1, 256, 128, 128, None)
    assert mq(1, None, None, None) == 0
    assert mq(5, None, None, None) == -22                                                          # unknown quantiser op
    assert mq(2, None, None, None) == -22                                                          # RMSNorm without its weight
    assert mq(3, p, p, p) == -22                                                                   # residual add in place
    assert mq(3, p, p, q) == 0
    assert mq(4, p, None, None, idx=p) == -22                                                      # SiLU x up takes no reorder index
    assert mq(1, None, None, None, M=3) == -33                                                     # one or two tokens only
    assert lib.atom_strerror(-33).startswith(b"unsupported")
    torch.cuda.synchronize()


def _attn_cfg(H=512, heads=4):
    return types.SimpleNamespace(hidden_size=H, num_attention_heads=heads, intermediate_size=1408, rms_norm_eps=1e-5,
                                 rope_theta=1e4)


def _load_layer(layer, seed):
    g = torch.Generator().manual_seed(seed)
    for mod in layer.modules():
        if type(mod).__name__ == "LinearInt4":
            w = (torch.randn(mod.out_features, mod.in_features, generator=g) * 0.05).half().cuda()
            mod.load_fp16_weight(w)
        elif type(mod).__name__ == "LlamaRMSNormInt4":
            mod.weight.data = (1 + 0.1 * torch.randn(mod.weight.shape, generator=g)).half().cuda()


def test_decoder_layer_prefill_then_decode_is_consistent():
    """The reference's real-kernel decoder layer (llama.py:247-292) on the HIP ops.  Consistency property that needs no
    external reference: decoding token S+1 against the INT4 cache written by a length-S prefill (append_kv_i4 +
    batch_decode_i4) must equal what a length-(S+1) PREFILL computes for its last token (init_kv_i4 + causal attention over
    the same de-quantised K/V) -- two disjoint code paths through the cache layout, RoPE and softmax."""
    from atom_amd import ops
    from atom_amd.e2e import LlamaDecoderLayer
    from atom_amd.utils import BatchLenInfo, BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4
    torch.manual_seed(3)
    cfg = _attn_cfg()
    dev = torch.device("cuda")
    layer = LlamaDecoderLayer(cfg, layer_idx=1).cuda()
    _load_layer(layer, 5)
    attn_in = []                                  # what the attention hands to its output projection
    orig_reorder = ops.reorder_fp16_i4

    def spy(a, idx, **kw):
        if idx is layer.self_attn.reorder_index:
            attn_in.append(a.clone())
        return orig_reorder(a, idx, **kw)
    lens = [37, 16]
    x = [(torch.randn(n + 1, cfg.hidden_size) * 0.7).half().cuda() for n in lens]       # S tokens + the decode token

    def fresh_pool():
        return KvPoolInt4(num_layers=2, num_heads=4, head_dim=128, capacity=16, block_len=16, device=dev)

    import atom_amd.e2e.llama as E
    fq0, E.FUSION.q_decode = E.FUSION.q_decode, False   # (the spy hooks the stand-alone reorder op: keep it a separate launch here)
    ops.reorder_fp16_i4 = spy
    # path A: prefill S, then one decode step
    pool = fresh_pool()
    cs = [KvCacheInt4(pool, n) for n in lens]
    y_pre = layer(torch.cat([xi[:-1] for xi in x]), BatchLenInfo(lens, 0, dev), BatchedKvCacheInt4(cs), None)
    assert y_pre.shape == (sum(lens), cfg.hidden_size) and torch.isfinite(y_pre).all()
    for c in cs:
        c.acquire_one()
    y_dec = layer(torch.stack([xi[-1] for xi in x]), BatchLenInfo([], len(lens), dev), None, BatchedKvCacheInt4(cs))
    # path B: prefill S+1 in one go
    pool2 = fresh_pool()
    cs2 = [KvCacheInt4(pool2, n + 1) for n in lens]
    y_full = layer(torch.cat(x), BatchLenInfo([n + 1 for n in lens], 0, dev), BatchedKvCacheInt4(cs2), None)
    ops.reorder_fp16_i4 = orig_reorder
    E.FUSION.q_decode = fq0
    rows = [lens[0], lens[0] + 1 + lens[1]]
    a_dec, a_full = attn_in[1].float(), attn_in[2][rows].float()
    assert (a_dec - a_full).abs().max().item() <= 2e-3 * a_full.abs().max().item()
    last = y_full[rows]
    # a W4A4 layer amplifies the M=2 vs M=55 kernels' last-bit differences through two more quantisers (a flipped INT4
    # code is a whole step): the sharp statement is on the attention output, the layer output gets the block-level bound
    err = (y_dec.float() - last.float()).abs().max().item()
    scale = last.float().abs().max().item()
    assert err <= 0.15 * scale, (err, scale)
    # both paths left the same codes in their caches (layer 1), modulo page numbering; the decode token went through
    # the M=2 decode GEMM (different FP32 summation order than the tile kernel), so a few of ITS codes may differ by one
    for c, c2 in zip(cs, cs2):
        a = torch.cat([pool.buf[i, 1] for i in c.indicies], dim=2)[:, :, :c.seqlen]
        b = torch.cat([pool2.buf[i, 1] for i in c2.indicies], dim=2)[:, :, :c2.seqlen]
        assert torch.equal(a[:, :, :-1], b[:, :, :-1])
        assert (a[:, :, -1] != b[:, :, -1]).float().mean().item() < 0.05


@pytest.mark.parametrize("M,N,K,nseg", [(1, 256, 640, 3), (5, 512, 1152, 2), (16, 4096, 4096, 3), (40, 1408, 640, 2), (16, 4096, 11008, 1), (64, 4096, 4096, 1),
                                        (7, 11008, 4096, 2)])
def test_multi_projection_launch_equals_separate_launches(M, N, K, nseg):
    """atom_gemm_w4a4_multi (decode batches: q / k / v, gate / up, down + residual in one launch): every segment bit-identical to
    the decode-batch kernel on that projection alone -- FP32 sums (atom_gemm_w4a4_f32), their fp16 rounding, and the residual add as
    torch adds two half tensors."""
    import types
    from atom_amd import ops
    from tests.helpers import rand_gemm_operands, to_device
    assert ops.multi_gemm_fits(M, N, nseg, K)
    ds = [rand_gemm_operands(M, N, K, seed=31 + i) for i in range(nseg)]
    devs = [to_device(d, "ref") for d in ds]
    a = devs[0]                                                   # the shared activation operand: a, a_scale, a_keeper, a_keeper_scale
    mods = [types.SimpleNamespace(weight_int4=torch.nn.Parameter(dv[1], requires_grad=False), weight_int8=torch.nn.Parameter(dv[5], requires_grad=False),
                                  scale_int4=torch.nn.Parameter(dv[3], requires_grad=False), scale_int8=torch.nn.Parameter(dv[7], requires_grad=False),
                                  packed=None) for dv in devs]
    for md in mods:
        md.packed = (lambda md=md: (md.weight_int4.data, md.weight_int8.data, md.scale_int4.data, md.scale_int8.data))
    want32 = [ops.dense_layer_gemm_i4_f32(a[0], dv[1], a[2], dv[3], a[4], dv[5], a[6], dv[7]) for dv in devs]
    fused = ops.fuse_projection_weights(mods)
    assert fused["key"] == ops.fused_key(mods)
    assert all(md.weight_int4.data_ptr() == fused["b4"].data_ptr() + i * N * (K - 128) // 2 for i, md in enumerate(mods))   # views, not copies
    res = (torch.randn((M, N), device="cuda") * 3).half()
    mask = 0b110 & ((1 << nseg) - 1)
    outs = ops.dense_layer_gemm_i4_multi(a[0], a[2], a[4], a[6], fused, f32_mask=mask, add=res)
    for i in range(nseg):
        if (mask >> i) & 1:
            assert outs[i].dtype == torch.float32 and torch.equal(outs[i], want32[i])
        elif i == 0:
            assert torch.equal(outs[0], want32[0].half() + res)
        else:
            assert torch.equal(outs[i], want32[i].half())
    outs = ops.dense_layer_gemm_i4_multi(a[0], a[2], a[4], a[6], fused)
    assert all(torch.equal(o, w.half()) for o, w in zip(outs, want32))
    if M >= 2:                                                    # (M = 1 goes to the dot-product kernel behind the plain entry point)
        assert torch.equal(outs[0], ops.dense_layer_gemm_i4_fp16(a[0], devs[0][1], a[2], devs[0][3], a[4], devs[0][5], a[6], devs[0][7]))


@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("op,N,K,nseg", [("rmsnorm", 4096, 4096, 3), ("reorder", 4096, 4096, 1), ("add_rmsnorm", 11008, 4096, 2), ("silu_mul", 4096, 11008, 1),
                                         ("rmsnorm", 1408, 640, 2), ("silu_mul", 512, 1408, 1), ("add_rmsnorm", 256, 5120, 3), ("reorder", 64, 256, 1),
                                         # (round 6: the dot-product kernel's ring depth follows a wave's feature steps -- 2, 4, and 8 steps on a
                                         # ring of 6: a second, refilling trip with a ragged last workgroup)
                                         ("reorder", 8192, 1024, 1), ("rmsnorm", 16384, 1024, 1), ("add_rmsnorm", 32000, 1024, 1)])
def test_quantiser_inside_the_gemm_launch_equals_quantiser_then_gemm(op, N, K, nseg, M):
    """atom_gemm_w4a4_multi_q (one or two tokens of a decode step): the quantiser that precedes the projections -- reorder, RMSNorm,
    residual add + RMSNorm, SiLU x up -- runs inside the GEMM launch, every workgroup on its own copy of the rows.  Every output
    (fp16, FP32 sums, fp16 + addend, the residual stream) bit-identical to the quantiser op followed by atom_gemm_w4a4_multi."""
    import types
    from atom_amd import ops
    from tests.helpers import rand_gemm_operands, to_device
    assert ops.multi_q_gemm_fits(op, M, N, nseg, K)
    devs = [to_device(rand_gemm_operands(4, N, K, seed=51 + i), "ref") for i in range(nseg)]
    mods = [types.SimpleNamespace(weight_int4=torch.nn.Parameter(dv[1], requires_grad=False), weight_int8=torch.nn.Parameter(dv[5], requires_grad=False),
                                  scale_int4=torch.nn.Parameter(dv[3], requires_grad=False), scale_int8=torch.nn.Parameter(dv[7], requires_grad=False),
                                  packed=None) for dv in devs]
    for md in mods:
        md.packed = (lambda md=md: (md.weight_int4.data, md.weight_int8.data, md.scale_int4.data, md.scale_int8.data))
    fused = ops.fuse_projection_weights(mods)
    g = torch.Generator(device="cuda").manual_seed(1000 * M + N + K)
    x = (torch.randn((M, K), device="cuda", generator=g) * 1.3).half()
    x2 = (torch.randn((M, K), device="cuda", generator=g) * 0.8).half()
    w = (1 + 0.1 * torch.randn(K, device="cuda", generator=g)).half()
    idx = torch.randperm(K, device="cuda", generator=g).to(torch.int16)
    res = (torch.randn((M, K), device="cuda", generator=g) * 2).half()
    add = (torch.randn((M, N), device="cuda", generator=g) * 3).half()
    eps = 1e-5
    # Round 6: for the token counts the SEPARATE entry points give to the dot-product kernel (one token; two with K > 4096:
    # atom_gemm_w4a4_packed_order == 64) the fused launch is that kernel with the quantiser in front (csrc/gemvq_w4a4.hip) -- same
    # summation order, compared directly.  Otherwise the fused launch runs the decode-batch kernel (the K items dealt to 8 waves) and
    # the separate ops are run on the tokens twice over so that they take that kernel as well (rows are independent).
    dot = ops.L.lib().atom_gemm_w4a4_packed_order(M, N * nseg, K, 0) == 64
    rep = 1 if dot else 2
    dup = lambda t: t.repeat(rep, 1)
    res_want = None
    if op == "reorder":
        qt = ops.reorder_fp16_i4(dup(x), idx)
        kw = dict(reorder_index=idx)
    elif op == "rmsnorm":
        qt = ops.rmsnorm_fp16_i4(dup(x), w, idx, eps)
        kw = dict(x2=w, reorder_index=idx, eps=eps)
    elif op == "add_rmsnorm":
        out = ops.add_rmsnorm_fp16_i4(dup(x), dup(res), w, idx, eps)
        res_want, qt = out[0][:M], out[1:]
        kw = dict(x2=w, residual=res, reorder_index=idx, eps=eps)
    else:
        qt = ops.activate_fp16_i4(dup(x), dup(x2))
        kw = dict(x2=x2)
    outlier, norms, outlier_scales, norm_scales = qt
    for mask, ad in ((0, None), (0b110 & ((1 << nseg) - 1), add)):
        want = ops.dense_layer_gemm_i4_multi(norms, norm_scales, outlier, outlier_scales, fused, f32_mask=mask, add=None if ad is None else dup(ad))
        want = [t[:M] for t in want]
        got, res_out = ops.dense_layer_gemm_i4_multi_q(op, x, fused, f32_mask=mask, add=ad, **kw)
        for i in range(nseg):
            assert got[i].dtype == want[i].dtype and torch.equal(got[i], want[i]), (op, M, i, mask)
        if op == "add_rmsnorm":
            assert torch.equal(res_out, res_want)
        else:
            assert res_out is None


@pytest.mark.parametrize("bsz", [1, 2])
def test_decoder_layer_quantisers_inside_the_gemms_equal_separate_launches(bsz):
    """A decode step of one or two tokens with the four quantisers inside their consumers' launches (7 launches) against the same
    step with separate quantiser launches (11): same cache contents, same output, bit for bit at two tokens (one token: see below)."""
    import atom_amd.e2e.llama as E
    from atom_amd.e2e import LlamaDecoderLayer
    from atom_amd.utils import BatchLenInfo, BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4
    torch.manual_seed(5)
    cfg = _attn_cfg()
    dev = torch.device("cuda")
    layer = LlamaDecoderLayer(cfg, layer_idx=0).cuda()
    _load_layer(layer, 8)
    ctx = 40
    x = (torch.randn(bsz, cfg.hidden_size) * 0.7).half().cuda()
    outs, caches = [], []
    mask0, mask20 = E.FUSION.q_mask, E.FUSION.q_mask2
    try:
        E.FUSION.q_mask = E.FUSION.q_mask2 = 15                # all four quantisers fused, at two tokens as well (default there: reorder -> o_proj only)
        for fq in (False, True):
            E.FUSION.q_decode = fq
            pool = KvPoolInt4(num_layers=1, num_heads=4, head_dim=128, capacity=bsz * 4, block_len=16, device=dev)
            g = torch.Generator(device="cuda").manual_seed(9)
            pool.buf.copy_(torch.randint(0, 255, pool.buf.shape, device=dev, dtype=torch.uint8, generator=g))
            pool.param.copy_((torch.rand(pool.param.shape, device=dev, generator=g) * 0.05 + 0.01).half())
            cs = [KvCacheInt4(pool, ctx) for _ in range(bsz)]
            for c in cs:
                c.acquire_one()
            outs.append(layer(x, BatchLenInfo([], bsz, dev), None, BatchedKvCacheInt4(cs)))
            caches.append((pool.buf.clone(), pool.param.clone()))
    finally:
        E.FUSION.q_decode = True
        E.FUSION.q_mask, E.FUSION.q_mask2 = mask0, mask20
    # (round 6: one token too -- the launches with the quantiser inside run the dot-product kernel's summation order, csrc/gemvq_w4a4.hip,
    # as the separate launches do; rounds 3-5 had the decode-batch kernel's order there and one fp16 ulp per projection between the two)
    assert torch.equal(caches[0][0], caches[1][0]) and torch.equal(caches[0][1].view(torch.int16), caches[1][1].view(torch.int16))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("bsz", [1, 3, 16])
def test_decoder_layer_fused_decode_step_equals_one_launch_per_projection(bsz):
    """A decode step of atom_amd.e2e.LlamaDecoderLayer with the round-3 launch fusions (q / k / v in one launch, gate / up in one
    launch, the second residual add inside down_proj's launch: 14 -> 10 launches) against the same step with one launch per
    projection (DecodeFusion.decode = False, the reference's call order llama.py:259-292): same cache contents and same output, bit for
    bit (batch 1: o_proj with its quantiser inside the launch runs the decode-batch kernel, o_proj alone the dot-product kernel --
    two summation orders, one fp16 ulp; every other projection of a one-token step takes the dot-product kernel on both sides)."""
    import atom_amd.e2e.llama as E
    from atom_amd.e2e import LlamaDecoderLayer
    from atom_amd.utils import BatchLenInfo, BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4
    torch.manual_seed(4)
    cfg = _attn_cfg()
    dev = torch.device("cuda")
    layer = LlamaDecoderLayer(cfg, layer_idx=0).cuda()
    _load_layer(layer, 7)
    ctx = 40
    x = (torch.randn(bsz, cfg.hidden_size) * 0.7).half().cuda()
    outs, caches = [], []
    for fused in (False, True):
        E.FUSION.decode = fused
        pool = KvPoolInt4(num_layers=1, num_heads=4, head_dim=128, capacity=bsz * 4, block_len=16, device=dev)
        g = torch.Generator(device="cuda").manual_seed(9)
        pool.buf.copy_(torch.randint(0, 255, pool.buf.shape, device=dev, dtype=torch.uint8, generator=g))
        pool.param.copy_((torch.rand(pool.param.shape, device=dev, generator=g) * 0.05 + 0.01).half())
        cs = [KvCacheInt4(pool, ctx) for _ in range(bsz)]
        for c in cs:
            c.acquire_one()
        outs.append(layer(x, BatchLenInfo([], bsz, dev), None, BatchedKvCacheInt4(cs)))
        caches.append((pool.buf.clone(), pool.param.clone()))
    E.FUSION.decode = True
    assert torch.equal(caches[0][0], caches[1][0]) and torch.equal(caches[0][1].view(torch.int16), caches[1][1].view(torch.int16))
    assert torch.equal(outs[0], outs[1])                      # (round 6: batch 1 included -- see csrc/gemvq_w4a4.hip)


@pytest.mark.parametrize("bsz,ctx", [(1, 520), (1, 130), (1, 1030), (1, 2500)])
def test_split_merge_inside_o_proj_equals_the_merge_launch(bsz, ctx):
    """Round 6: with the KV range of a decode step split over several waves (long contexts at small batches), the decode op leaves its
    partial states un-merged and o_proj's launch merges them in front of its reorder quantiser (atom_gemm_w4a4_multi_merge_q).  The op
    against batch_decode_i4 (merge launch) -> dense_layer_gemm_i4_multi_q("reorder"), and a whole decode layer with and without it:
    bit for bit (waves in workgroups of four that leave one partial state each, or single waves: 2 / 2 / 4 / 8 partial states here)."""
    import atom_amd.e2e.llama as E
    from atom_amd import ops
    from atom_amd.e2e import LlamaDecoderLayer
    from atom_amd.utils import BatchLenInfo, BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4
    torch.manual_seed(11)
    cfg = _attn_cfg()
    dev = torch.device("cuda")
    layer = LlamaDecoderLayer(cfg, layer_idx=0).cuda()
    _load_layer(layer, 13)
    x = (torch.randn(bsz, cfg.hidden_size) * 0.7).half().cuda()
    outs, caches = [], []
    nblk = bsz * (ctx // 16 + 2)
    prev = E.FUSION.merge_in_o_proj
    for fused in (False, True):
        E.FUSION.merge_in_o_proj = fused
        try:
            pool = KvPoolInt4(num_layers=1, num_heads=4, head_dim=128, capacity=nblk, block_len=16, device=dev)
            g = torch.Generator(device="cuda").manual_seed(9)
            pool.buf.copy_(torch.randint(0, 255, pool.buf.shape, device=dev, dtype=torch.uint8, generator=g))
            pool.param.copy_((torch.rand(pool.param.shape, device=dev, generator=g) * 0.05 + 0.01).half())
            pool._free = set(range(nblk))
            cs = [KvCacheInt4(pool, ctx) for _ in range(bsz)]
            for c in cs:
                c.acquire_one()
            kv = BatchedKvCacheInt4(cs)
            if fused:
                splits = ops.decode_splits(bsz, kv)
                assert splits >= 2 and ops.merge_q_gemm_fits(bsz, cfg.hidden_size, 1, cfg.hidden_size, splits), splits
            outs.append(layer(x, BatchLenInfo([], bsz, dev), None, kv))
            caches.append((pool.buf.clone(), pool.param.clone()))
        finally:
            E.FUSION.merge_in_o_proj = prev
    assert torch.equal(caches[0][0], caches[1][0]) and torch.equal(caches[0][1].view(torch.int16), caches[1][1].view(torch.int16))
    assert torch.equal(outs[0], outs[1])
    # the op itself, on the partial states of a decode over the cache just written
    at = layer.self_attn
    q = (torch.randn(bsz, 4, 128, device=dev)).half()
    part = ops.batch_decode_i4(q, kv, 0, merge=False)
    o = ops.batch_decode_i4(q, kv, 0)
    add = (torch.randn(bsz, cfg.hidden_size, device=dev) * 2).half()
    (want,), _ = ops.dense_layer_gemm_i4_multi_q("reorder", o.view(bsz, -1), at.o_proj.single(), reorder_index=at.reorder_index, add=add)
    (got,) = ops.dense_layer_gemm_i4_merge_q(part, ops.decode_splits(bsz, kv), at.o_proj.single(), reorder_index=at.reorder_index, add=add)
    assert torch.equal(got, want)


@pytest.mark.parametrize("bsz,heads,ctx,n_out", [(1, 32, 1030, 1024), (1, 32, 2500, 1024), (1, 16, 1030, 1024), (1, 40, 1030, 1024), (1, 8, 1030, 1024),
                                                   (1, 32, 1030, 8192), (1, 32, 520, 4096)])
def test_merge_q_op_at_llama_widths(bsz, heads, ctx, n_out):
    """The op alone at Llama-7B / 13B widths (hidden 4096: every thread of the eight quantiser waves holds a chunk of the row; 5120: the
    row needs more than eight waves, no roles), 2-16 partial states per head (two tokens never reach the op: they run the dot-product kernel
    only for K > 4096, and two rows of that width exceed its one chunk per thread): merge_q on the partial states of a decode
    against the decode op's own merge launch -> multi_q("reorder"), bit for bit.  8192 outputs: more than two feature steps per streamer
    wave, so the quantiser waves stream features too (roles without the owning streamers); 4096: the decode layer's own shape."""
    from atom_amd import ops
    from atom_amd.e2e.llama import LinearInt4
    from atom_amd.utils import BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4
    dev = torch.device("cuda")
    hs = heads * 128
    g = torch.Generator(device="cuda").manual_seed(heads + ctx)
    lin = LinearInt4(hs, n_out, out_dtype="fp16", bias=False).cuda()
    lin.load_fp16_weight((torch.randn(n_out, hs, device=dev, generator=g) * 0.05).half())
    idx = torch.randperm(hs, device=dev, generator=g).to(torch.int16)
    nblk = bsz * (ctx // 16 + 2)
    pool = KvPoolInt4(num_layers=1, num_heads=heads, head_dim=128, capacity=nblk, block_len=16, device=dev)
    pool.buf.copy_(torch.randint(0, 255, pool.buf.shape, device=dev, dtype=torch.uint8, generator=g))
    pool.param.copy_((torch.rand(pool.param.shape, device=dev, generator=g) * 0.05 + 0.01).half())
    pool._free = set(range(nblk))
    kv = BatchedKvCacheInt4([KvCacheInt4(pool, ctx) for _ in range(bsz)])
    splits = ops.decode_splits(bsz, kv)
    if splits < 2 or not ops.merge_q_gemm_fits(bsz, n_out, 1, hs, splits):
        pytest.skip(f"{splits} partial states / shape outside the merge op")
    q = torch.randn((bsz, heads, 128), device=dev, generator=g).half()
    part = ops.batch_decode_i4(q, kv, 0, merge=False)
    o = ops.batch_decode_i4(q, kv, 0)
    (want,), _ = ops.dense_layer_gemm_i4_multi_q("reorder", o.view(bsz, hs), lin.single(), reorder_index=idx)
    (got,) = ops.dense_layer_gemm_i4_merge_q(part, splits, lin.single(), reorder_index=idx)
    assert torch.equal(got, want)


def test_fused_quantiser_query_is_the_launchers_predicate_at_wide_hidden_sizes():
    """Round-3 advisor item: the shape query of atom_gemm_w4a4_multi_q and its launchers share ONE predicate -- what the query accepts
    launches, what it refuses raises (it used to say yes to shapes the launch then failed on), and a decode layer asks it per batch size.
    Round 6: one and two tokens run the quantiser in front of the dot-product kernel (csrc/gemvq_w4a4.hip: 1024 threads stage the rows,
    so hidden 8192 now fits at two tokens as well); the reorder / RMSNorm variants stop at two tokens of hidden > 8192 (two 16-byte
    row chunks per thread), SiLU x up at two tokens of more than 16,384 channels."""
    import types
    import atom_amd.e2e.llama as E
    from atom_amd import ops
    from atom_amd._lib import AtomHipError
    from atom_amd.e2e import LlamaDecoderLayer
    from atom_amd.utils import BatchLenInfo, BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4
    from tests.helpers import rand_gemm_operands, to_device
    assert ops.multi_q_gemm_fits("reorder", 1, 8192, 1, 8192) and ops.multi_q_gemm_fits("reorder", 2, 8192, 1, 8192)
    assert ops.multi_q_gemm_fits("rmsnorm", 2, 6656, 3, 6656) and ops.multi_q_gemm_fits("silu_mul", 2, 4096, 1, 8192)
    assert not ops.multi_q_gemm_fits("reorder", 2, 4096, 1, 11008) and ops.multi_q_gemm_fits("silu_mul", 2, 4096, 1, 11008)
    assert ops.multi_q_gemm_fits("reorder", 1, 64, 1, 12288) and not ops.multi_q_gemm_fits("reorder", 2, 64, 1, 12288)
    dev = torch.device("cuda")
    # what the query refuses raises: two tokens of 11008 channels through the reorder quantiser
    N, K = 64, 11008
    dv = to_device(rand_gemm_operands(2, N, K, seed=5), "ref")
    md = types.SimpleNamespace(weight_int4=torch.nn.Parameter(dv[1], requires_grad=False), weight_int8=torch.nn.Parameter(dv[5], requires_grad=False),
                               scale_int4=torch.nn.Parameter(dv[3], requires_grad=False), scale_int8=torch.nn.Parameter(dv[7], requires_grad=False))
    md.packed = lambda: (md.weight_int4.data, md.weight_int8.data, md.scale_int4.data, md.scale_int8.data)
    fused = ops.fuse_projection_weights([md])
    x2 = (torch.randn(2, K) * 0.7).half().cuda()
    idx = torch.randperm(K).to(torch.int16).cuda()
    with pytest.raises(AtomHipError):
        ops.dense_layer_gemm_i4_multi_q("reorder", x2, fused, reorder_index=idx)
    ops.dense_layer_gemm_i4_multi_q("reorder", x2[:1].contiguous(), fused, reorder_index=idx)       # one token fits
    # a decode layer at hidden 8192, two tokens: fused (it fits since round 6) == separate launches, bit for bit
    cfg = _attn_cfg(H=8192, heads=64)
    layer = LlamaDecoderLayer(cfg, layer_idx=0).cuda()
    _load_layer(layer, 21)
    x = (torch.randn(2, 8192) * 0.7).half().cuda()
    outs = []
    for fq in (True, False):
        E.FUSION.q_decode = fq
        try:
            pool = KvPoolInt4(num_layers=1, num_heads=64, head_dim=128, capacity=8, block_len=16, device=dev)
            g = torch.Generator(device="cuda").manual_seed(9)
            pool.buf.copy_(torch.randint(0, 255, pool.buf.shape, device=dev, dtype=torch.uint8, generator=g))
            pool.param.copy_((torch.rand(pool.param.shape, device=dev, generator=g) * 0.05 + 0.01).half())
            cs = [KvCacheInt4(pool, 20) for _ in range(2)]
            for c in cs:
                c.acquire_one()
            outs.append(layer(x, BatchLenInfo([], 2, dev), None, BatchedKvCacheInt4(cs)))
        finally:
            E.FUSION.q_decode = True
    assert layer._fused_q_fits(2) is True and layer._fused_q_fits(1) is True
    assert torch.equal(outs[0], outs[1])
