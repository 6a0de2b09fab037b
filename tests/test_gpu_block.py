"""BASELINE config 4 family: one W4A4 Llama decoder block through the drop-in operator surface (atom_amd.model.*)
on the GPU, against the output of the UNMODIFIED reference QLlamaDecoderLayer (tests/golden/gen_golden_block.py).
Every GEMM of the block runs on atom_gemm_w4a4_f16 and every activation quantisation on the fused HIP kernels."""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _args(**over):
    d = dict(wbits=4, abits=4, a_sym=True, w_sym=True, act_group_size=128, weight_group_size=128,
             weight_channel_group=2, keeper=128, keeper_precision=3, a_clip_ratio=0.9, w_clip_ratio=0.85,
             kv_clip_ratio=1.0, tiling=0, exponential=False, quant_type="int", static=False, reorder=True, kv_cache=True)
    d.update(over)
    return types.SimpleNamespace(**d)


def _build(device):
    import gen_golden_block as G          # only its pure-torch builders; it does not touch /root/reference here
    from atom_amd.model import quant, qLlamaLayer
    hidden, heads, inter, bsz, seq = 512, 4, 1408, 2, 24
    args = _args()
    orig = G.build_original(hidden, heads, inter, seed=7)
    wsum = 0.0
    for mod in (orig.self_attn.q_proj, orig.self_attn.k_proj, orig.self_attn.v_proj, orig.self_attn.o_proj,
                orig.mlp.gate_proj, orig.mlp.up_proj, orig.mlp.down_proj):
        wsum += float(mod.weight.detach().double().abs().sum())
    wsum += float(orig.input_layernorm.weight.detach().double().abs().sum() + orig.post_attention_layernorm.weight.detach().double().abs().sum())
    idx, x, pos, mask = G.make_inputs(hidden, inter, bsz, seq, seed=8)
    m = qLlamaLayer.QLlamaDecoderLayer(orig, args).to(device)
    idx = {k: v.to(device) for k, v in idx.items()}
    G.prepare(m, args, idx, quant)
    return m, x, pos, mask, wsum


def test_llama_block_matches_reference(golden_dir):
    """Three statements, from sharp to statistical:
    (1) teacher-forced, per layer: with the SAME input every one of the 7 W4A4 GEMMs agrees with F.linear on the
        fake-quant operands (the reference's forward, qLinearLayer.py:32-35) within the north-star 1e-2;
    (2) the block evaluated through our modules with the packed weights dropped (fused HIP quant kernels + F.linear)
        reproduces the reference's output to 5e-3: reorder / RoPE / attention / residual wiring is right;
    (3) the full HIP path end to end, measured where the difference is made: at every quantiser that feeds a GEMM the INT4
        codes of the HIP run are compared with those of the reference-order run (2).  A 1e-3 perturbation of a GEMM output
        (integer-exact vs fp16-rounded fake-quant operands) flips a fraction of a percent of the next quantiser's codes, each
        flip is a whole quantisation step, and the flips compound through the block -- so the bound is on the flip rate per
        quantiser (0 for q/k/v, < 2-3 % one GEMM downstream, < 12 % at down_proj, which sees every upstream flip through
        SiLU x up; measured 0 / 0.7 % / 7 % -- a GEMM wired to a wrong scale flips > 30 % at the next quantiser), and the
        end-to-end deviation that follows from it."""
    z = np.load(os.path.join(golden_dir, "llama_block_512.npz"))
    m, x, pos, mask, wsum = _build("cuda")
    assert np.array_equal(x.numpy(), z["x"]) and abs(wsum - float(z["weight_abs_sum"])) < 1e-6 * wsum
    from atom_amd.model import quant
    from atom_amd.model.qLinearLayer import find_qlinear_layers
    layers = find_qlinear_layers(m)
    assert len(layers) == 7 and all(l.packed_weight() is not None for l in layers.values())
    seen, codes_hip, codes_ref = {}, {}, {}

    def hook(name, store, check):
        def f(mod, inp, out):
            cd = quant.get_codes(inp[0])
            assert cd is not None, f"{name}: activation codes lost -> would silently use F.linear"
            assert not cd.wide                                   # 48 tokens: packed nibbles
            store[name] = (cd.o4.clone(), cd.o8.clone())
            if check:
                ref = torch.nn.functional.linear(inp[0], mod.weight, mod.bias).float()
                seen[name] = ((out.float() - ref).abs().max() / ref.pow(2).mean().sqrt()).item()
        return f
    handles = [l.register_forward_hook(hook(n_, codes_hip, True)) for n_, l in layers.items()]
    y = m(x.cuda(), attention_mask=mask.cuda(), position_ids=pos.cuda())[0]
    for h in handles:
        h.remove()
    assert len(seen) == 7 and max(seen.values()) <= 1e-2, seen                          # (1)
    want = z["y"].astype(np.float64)
    got = y.float().cpu().numpy().astype(np.float64)
    for l in layers.values():                    # force the reference forward (F.linear on the fake-quant weight)
        l._packed = None
        l._unpackable_key = l._weight_key()
    handles = [l.register_forward_hook(hook(n_, codes_ref, False)) for n_, l in layers.items()]
    y2 = m(x.cuda(), attention_mask=mask.cuda(), position_ids=pos.cuda())[0].float().cpu().numpy().astype(np.float64)
    for h in handles:
        h.remove()
    assert np.linalg.norm(y2 - want) / np.linalg.norm(want) < 5e-3                      # (2)
    assert np.abs(y2 - want).max() <= 0.1 * np.sqrt((want ** 2).mean())
    flips = {}
    for n_ in layers:                                                                   # (3)
        a4, a8 = codes_hip[n_]
        b4, b8 = codes_ref[n_]
        a4, b4 = a4.view(torch.uint8), b4.view(torch.uint8)
        nib = ((a4 & 0xF) != (b4 & 0xF)).float().mean() + ((a4 >> 4) != (b4 >> 4)).float().mean()
        flips[n_] = (0.5 * nib.item(), (a8 != b8).float().mean().item())
    print("code flips per quantiser (INT4, INT8 keeper):", {k: (round(v[0], 5), round(v[1], 5)) for k, v in flips.items()})
    limit = {"self_attn.q_proj": 0.0, "self_attn.k_proj": 0.0, "self_attn.v_proj": 0.0,      # same block input: no flips at all
             "self_attn.o_proj": 0.03, "mlp.gate_proj": 0.02, "mlp.up_proj": 0.02,            # one GEMM (+ attention) upstream
             "mlp.down_proj": 0.12}                                                           # everything upstream, through SiLU x up
    for n_, (f4, _) in flips.items():
        assert f4 <= limit[n_], (n_, flips)
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    print("end-to-end relative Frobenius deviation of the HIP run:", rel)
    assert rel < 0.10


def test_fused_layers_match_unfused_reference_order():
    """QLlamaRMSNorm / QLlamaMLP fused HIP paths vs the reference's op-by-op order evaluated with torch on the GPU."""
    from functools import partial
    from atom_amd.model import quant, qLlamaLayer
    from transformers.models.llama.modeling_llama import LlamaRMSNorm
    args = _args()
    torch.manual_seed(0)
    H = 1024
    norm = LlamaRMSNorm(H, eps=1e-5)
    norm.weight.data = 1 + 0.1 * torch.randn(H)
    norm = norm.half().cuda()
    qn = qLlamaLayer.QLlamaRMSNorm(norm, args).cuda()
    idx = torch.randperm(H).cuda()
    qn.register_buffer("reorder_index", idx)
    qn.act_quant.configure(partial(quant.quantize_activation_wrapper, args=args), None)
    x = (torch.randn(3, 5, H) * 2).half().cuda()
    fused = qn(x)
    assert quant.get_codes(fused) is not None
    ref = torch.index_select(norm(x), 2, idx)
    # torch-op restatement of the reference wrapper (non-hot path of our quant.py uses the same formulas)
    a2 = _args(keeper_precision=3)
    keep = quant.quantize_tensor(ref.reshape(-1, H)[:, -128:].clone(), n_bits=8, group_size=0, tiling=0, sym=True)
    body = ref.reshape(-1, H).clone(); body[:, -128:] = 0
    body = quant.quantize_tensor(body, n_bits=4, group_size=128, tiling=0, sym=True, clip_ratio=0.9)
    body[:, -128:] = keep
    # torch.rsqrt on the GPU vs our correctly rounded 1/sqrt: a last-bit difference in one group's absmax moves that
    # group's scale by one fp16 ulp (all 128 values change in the last bit); bit-exactness of the kernel itself against
    # its specification is test_gpu_quant.py::test_rmsnorm_quant_bit_exact.
    f, b = fused.reshape(-1, H).float(), body.float()
    assert (f == b).float().mean().item() > 0.9
    assert ((f - b).norm() / b.norm()).item() < 2e-2


# BASELINE config 4 at its stated size: Llama-7B projections (hidden 4096, intermediate 11008) on 4096 tokens
LLAMA7B = [("q/k/v/o_proj", 4096, 4096, "reorder"), ("gate/up_proj", 11008, 4096, "reorder"), ("down_proj", 4096, 11008, "silu")]


@pytest.mark.parametrize("name,N,K,feeder", LLAMA7B)
def test_llama7b_projection_teacher_forced(name, N, K, feeder):
    """Teacher-forced parity of one W4A4 projection at Llama-7B size, M = 4096 tokens, through the operator surface
    (QLinearLayer + the fused quantiser that feeds it in the block): the reference forward is F.linear on the fake-quant fp16
    operands (model/qLinearLayer.py:32-35) -- evaluated here with torch on the same device on the SAME fake-quant tensors the
    HIP quantiser / packer produced (those are pinned bit-exactly against reference goldens in test_gpu_quant.py).
      (1) max |y - ref| <= 1e-2 rms(ref)  (north star);
      (2) what the difference does downstream: both outputs go through the next activation quantiser (RMSNorm-free reorder
          quant, clip 0.9) and the INT4 codes that differ are counted -- a flip is one whole quantisation step, so this
          rate is what bounds the error a block accumulates (the old 10 % end-to-end bound could not see a mis-scaled GEMM)."""
    from functools import partial
    from atom_amd import ops
    from atom_amd.model import quant
    from atom_amd.model.qLinearLayer import QLinearLayer
    args = _args()
    M = 4096
    g = torch.Generator(device="cuda").manual_seed(N + K)
    lin = torch.nn.Linear(K, N, bias=False)
    lin.weight.data = (torch.randn((N, K), device="cuda", generator=g) * 0.02).half()
    layer = QLinearLayer(lin.half().cuda(), args)
    layer.quant()
    x = (torch.randn((M, K), device="cuda", generator=g) * 1.0).half()
    x[:, -128:] *= 15                                           # outlier channels sit in the keeper after the reorder
    if feeder == "silu":
        up = (torch.randn((M, K), device="cuda", generator=g)).half()
        o = ops.activate_fp16_i4(x, up, quant_mode="sim", clip=0.9, scale_layout="plain", return_dequant=True,
                                 wide_codes=quant.want_wide_codes(M))
    else:
        o = ops.reorder_fp16_i4(x, None, quant_mode="sim", clip=0.9, scale_layout="plain", return_dequant=True,
                                wide_codes=quant.want_wide_codes(M))
    xq = quant.attach_codes(o[4], quant.ActCodes(o[0], o[1], o[2], o[3], M, K, wide=quant.want_wide_codes(M)))
    y = layer(xq)
    assert layer._f6 is not None                                # the F6 route (256x256 kernel at this size) was taken
    ref = torch.nn.functional.linear(o[4], layer.weight)
    rms = ref.float().pow(2).mean().sqrt()
    err = (y.float() - ref.float()).abs().max() / rms
    assert err.item() <= 1e-2, (name, err.item())
    qa = ops.reorder_fp16_i4(y, None, quant_mode="sim", clip=0.9, scale_layout="plain")
    qb = ops.reorder_fp16_i4(ref, None, quant_mode="sim", clip=0.9, scale_layout="plain")
    na = torch.stack([qa[1] & 0xF, qa[1] >> 4], -1)
    nb = torch.stack([qb[1] & 0xF, qb[1] >> 4], -1)
    flips = (na != nb).float().mean().item()
    assert flips <= 5e-3, (name, flips)                         # measured 1-3e-3 (profiles/r02_block_llama7b.txt)


WIDE_PROJ = {   # projection -> (fixture key of its input, module path)
    "q": ("xq1", "self_attn.q_proj"), "k": ("xq1", "self_attn.k_proj"), "v": ("xq1", "self_attn.v_proj"), "o": ("attn_q", "self_attn.o_proj"),
    "gate": ("xq2", "mlp.gate_proj"), "up": ("xq2", "mlp.up_proj"), "down": ("act_q", "mlp.down_proj"),
}


def test_llama7b_projections_teacher_forced_on_256_reference_rows(golden_dir):
    """The WIDE teacher-forcing sample (tests/golden/llama_block_7b_wide.npz, gen_golden_block7b.py): for 256 token rows of the
    unmodified reference's QLlamaDecoderLayer.forward at hidden 4096 / intermediate 11008, every one of the seven W4A4 GEMMs is fed
    the REFERENCE's own input -- its fake-quantised activation tensor in the integer view the fixture's recovered per-group scales
    give (codes = value / scale, exact) -- through atom_gemm_w4a4 with the HIP-quantised weights, and compared with the REFERENCE's
    own output on 256 sampled features: 256 x 256 elements per projection, where a 5 % systematic error in one projection would be
    two orders of magnitude above the bound (the 16-row sample of the test below could not see it).  The deviation that remains is
    the reference's own: its F.linear runs on fp16-ROUNDED fake-quant operands and rounds the sum once more (3.4e-4 relative; the
    CPU oracle pinned on the same fixture shows the same number, tests/test_oracle_block7b.py), so the HIP output is ALSO compared
    with the C restatement of the contract on the same integer operands, in the summation order of the kernel the shape is dispatched
    to (atom_gemm_w4a4_ws_recodes / atom_gemm_w4a4_f6_order): bit for bit.
    The first quantiser (input_layernorm -> reorder -> quantise) is compared on all 256 rows as well."""
    import gen_golden_block7b as G7
    import gen_golden_block as G
    import c_oracle as C
    from atom_amd import ops, _lib
    from atom_amd.model import quant, qLlamaLayer
    from atom_amd.model.qLinearLayer import find_qlinear_layers
    lib = _lib.lib()
    z = np.load(os.path.join(golden_dir, "llama_block_7b_wide.npz"))
    args = _args()
    orig = G7.build_original()
    idx, x, _, _ = G7.make_inputs()
    m = qLlamaLayer.QLlamaDecoderLayer(orig, args).to("cuda")
    G.prepare(m, args, {k: v.cuda() for k, v in idx.items()}, quant)
    layers = find_qlinear_layers(m)
    wr = torch.from_numpy(z["rows"]).cuda()
    xq1 = m.input_layernorm(x.cuda()[:, wr])[0]
    differ = (xq1 != torch.from_numpy(z["xq1"]).cuda()).float().mean().item()
    print("input_layernorm -> quant on 256 reference rows: fraction of differing elements", differ)
    assert differ <= 5e-3
    report = {}
    for name, (key, path) in WIDE_PROJ.items():
        v = torch.from_numpy(z[key]).cuda().float()
        s4 = torch.from_numpy(z["s4_" + key]).cuda()
        s8 = torch.from_numpy(z["s8_" + key]).cuda()
        M, H = v.shape
        c4 = torch.round(v[:, :-128].reshape(M, -1, 128) / s4.float()[..., None]).reshape(M, H - 128).to(torch.int32)
        c8 = torch.round(v[:, -128:] / s8.float()[:, None]).to(torch.int8)
        assert int(c4.min()) >= -8 and int(c4.max()) <= 7
        o4 = ((c4[:, 0::2] & 0xF) | ((c4[:, 1::2] & 0xF) << 4)).to(torch.uint8).view(torch.int8).contiguous()
        sA = s4.t().contiguous()
        b4, b8, sb, sb8 = layers[path].packed_weight()
        D = ops.dense_layer_gemm_i4_fp16(o4, b4, sA, sb, c8.contiguous(), b8, s8, sb8, scale_layout="plain")
        cols = torch.from_numpy(z["cols_" + name]).cuda()
        got = D[:, cols].double()
        ref = torch.from_numpy(z["out_" + name]).cuda().double()
        rel = ((got - ref).norm() / ref.norm()).item()
        worst = ((got - ref).abs() / ref.abs().clamp_min(ref.pow(2).mean().sqrt())).max().item()
        # ... and against the C restatement of the contract on the same integer operands (the sampled features of the HIP weights)
        # (in the summation order of the kernel this shape is dispatched to: the library's own queries)
        N, K = b4.shape[0], H
        order = lib.atom_gemm_w4a4_packed_order(M, N, K, 2)               # (ops keeps every weight's BF6 form) 1: K steps in order (tile kernels), 2 / 4: ordered K ranges (BF6
        assert order in (1, 2, 4, 8), order                               # K-group kernels), 8: the decode-batch kernel
        nsplit = -order if order in (2, 4) else order
        Dc = C.gemm(o4.cpu().numpy().view(np.uint8), b4[cols].cpu().numpy().view(np.uint8), sA.cpu().numpy(),
                    sb.reshape(sA.shape[0], -1)[:, cols].cpu().numpy(), c8.cpu().numpy(), b8[cols].cpu().numpy(), s8.cpu().numpy(),
                    sb8.reshape(-1)[cols].cpu().numpy(), nsplit=nsplit)
        dc = torch.from_numpy(Dc.astype(np.float64)).cuda()
        ndiff = (got != dc).float().mean().item()
        report[name] = (rel, worst, nsplit, ndiff)
        assert rel <= 1e-3 and worst <= 1e-2, (name, rel, worst)
        assert ndiff == 0.0, (name, nsplit, ndiff)
    print("HIP GEMM on the reference's input vs the reference's output, 256 rows x 256 features per projection "
          "(relative Frobenius, worst element / max(|ref|, rms)) and vs the C contract (summation order, fraction of elements that differ):")
    for k, r in report.items():
        print(f"  {k:5s} {r[0]:.2e} {r[1]:.2e} | {r[2]} {r[3]:.4f}")


def test_qlinear_weight_memory_policy():
    """keep_packed_with_f6 = False: after the first prefill batch the layer holds the F6 form (6.75 bit per weight incl. its fp32
    scales) and no packed INT4 codes; a decode-size batch afterwards re-packs them from the fp16 weight; results are those of a layer
    that keeps both forms."""
    from atom_amd.model import quant
    from atom_amd.model.qLinearLayer import QLinearLayer
    args = _args()
    torch.manual_seed(2)
    N, K = 1024, 2176
    mk = lambda: QLinearLayer(torch.nn.Linear(K, N, bias=False).half().cuda(), args)
    a, b = mk(), mk()
    w0 = a.weight.clone()
    b.weight = w0.clone()
    b.keep_packed_with_f6 = False
    a.quant(); b.quant()
    big = quant.hip_act_quant((torch.randn(300, K, device="cuda") * 1.5).half(), args)
    small = quant.hip_act_quant((torch.randn(5, K, device="cuda") * 1.5).half(), args)
    assert torch.equal(a(big), b(big))
    wa, wb = a.weight_bytes(), b.weight_bytes()
    assert wa["packed_int4"] == N * (K - 128) // 2 and wb["packed_int4"] == 0
    assert wb["f6"] == wa["f6"] == (K // 128 - 1) * N * 108
    assert 8.0 * (wb["f6"] + wb["keeper_and_scales"]) / (N * K) < 7.8      # bits per weight without the packed codes (keeper incl.)
    assert torch.equal(a(small), b(small))                                  # re-packed on demand
    # keep_f6 = False (round 6): 4-bit weights only -- the BF6 form is transient; same bits, no F6 bytes held
    n4 = mk()
    n4.weight = w0.clone()
    n4.keep_f6 = False
    n4.quant()
    assert torch.equal(n4(big), a(big)) and torch.equal(n4(small), a(small)) and torch.equal(n4(big), a(big))
    assert n4.weight_bytes()["f6"] == 0 and n4.weight_bytes()["packed_int4"] == wa["packed_int4"]
    assert b.weight_bytes()["packed_int4"] == wa["packed_int4"]
    assert torch.equal(a(big), b(big))
    # a .to() round trip with released codes (None entries in the packed tuple) and a cached F6 form
    c = mk()
    c.weight = w0.clone()
    c.keep_packed_with_f6 = False
    c.quant()
    want = a(big)
    assert torch.equal(c(big), want) and c.weight_bytes()["packed_int4"] == 0
    c = c.to("cuda").half()
    assert torch.equal(c(big), want)
    c = c.to("cpu").to("cuda")
    assert torch.equal(c(big), want) and torch.equal(c(small), a(small))


def test_mlp_fused_gate_up_honours_weight_memory_policy():
    """QLlamaMLP builds its fused gate/up operand from the packed INT4 codes of both layers; with keep_packed_with_f6 = False the
    codes are re-packed for the build only and released again, and later batches (cache hit) do not bring them back."""
    import gen_golden_block as G
    from atom_amd.model import quant, qLlamaLayer
    args = _args()
    hidden, inter = 1024, 2816
    orig = G.build_original(hidden, 8, inter, seed=11)
    mlp = qLlamaLayer.QLlamaMLP(orig.mlp, args).to("cuda")
    ref = qLlamaLayer.QLlamaMLP(G.build_original(hidden, 8, inter, seed=11).mlp, args).to("cuda")
    for mod in (mlp, ref):
        from functools import partial
        mod.act_quant.configure(partial(quant.quantize_activation_wrapper, args=args), None)
        for l in (mod.gate_proj, mod.up_proj, mod.down_proj):
            l.quant()
    for l in (mlp.gate_proj, mlp.up_proj, mlp.down_proj):
        l.keep_packed_with_f6 = False
    x = quant.hip_act_quant((torch.randn(4096, hidden, device="cuda") * 1.5).half(), args)
    assert quant.get_codes(x).wide == "f6"
    assert torch.equal(mlp(x), ref(x)) and mlp._fused is not None
    held = lambda mod: sum(l.weight_bytes()["packed_int4"] for l in (mod.gate_proj, mod.up_proj, mod.down_proj))
    assert held(ref) == 3 * hidden * inter // 2 - 128 * inter - 64 * hidden and held(mlp) == 0
    assert torch.equal(mlp(x), ref(x)) and held(mlp) == 0


def test_llama7b_block_matches_reference_at_stated_width(golden_dir):
    """BASELINE config 4 at its stated WIDTH (hidden 4096 = 32 heads x 128, intermediate 11008), batch 1 x 2048 tokens -- the batch
    the CPU oracle can hold -- against the UNMODIFIED reference QLlamaDecoderLayer.forward (model/qLlamaLayer.py:86-127)
    run on CPU by tests/golden/gen_golden_block7b.py.  Weights, indices and x are re-generated from the same seeds (checksums in
    the fixture); the fixture holds 79 sampled token rows of the block output y and of the MLP half's input h, 16 rows of the
    tensors either side of every quantiser, and per-row / whole-tensor checksums of y.
      (1) teacher-forced on the device: every one of the 7 W4A4 GEMMs (F6 operands on the block-scaled MFMA kernels, 2048 rows)
          within 1e-2 max(|ref|, rms) of F.linear on the same fake-quant operands;
      (2) teacher-forced against the REFERENCE's intermediates, stage by stage on the stored rows (every row-wise stage is fed the
          reference's own input for it): RMSNorm-quantisers and the SiLU x up quantiser differ from the reference's tensors on at
          most the stated fraction of elements (one code step; the documented RMSNorm / expf tolerance), quantiser + GEMM stages
          within the error those flips explain;
      (3) the block as a whole.  A W4A4 block is CHAOTIC in its rounding: a perturbation d of a quantiser's input flips a
          fraction ~ d / step of its codes by a whole step, so the error behind the quantiser is ~ sqrt(d * step), not d.  The
          integer-exact HIP GEMM differs from F.linear on the fp16-ROUNDED fake-quant operands (half(code * s) is not code * s) by
          3.4e-4 relative -- 1/30 of the north-star tolerance -- and torch's own F.linear on this GPU differs from torch's F.linear
          on the CPU by ~1e-5; at the block output these become 0.18 and 0.12.  So the end-to-end statement is calibrated, not
          absolute: the reference-order run (our modules with the packed weights dropped: HIP quantisers + F.linear) is repeated
          with every GEMM output perturbed by Gaussian noise of the Frobenius size the HIP GEMM's own deviation has (measured in
          (1)), and the HIP run must be no further from the reference golden than that perturbed reference run is (+ 25 %); plus
          INT4 code flips per quantiser between the HIP and the reference-order run, the attention half, row checksums, and a
          hard cap that a mis-wired scale or index breaks (> 0.5)."""
    import gen_golden_block7b as G7
    import gen_golden_block as G
    from atom_amd import ops
    from atom_amd.model import quant, qLlamaLayer
    from atom_amd.model.qLinearLayer import find_qlinear_layers
    z = np.load(os.path.join(golden_dir, "llama_block_7b_1x2048.npz"))
    args = _args()
    orig = G7.build_original()
    wsum = G7.weight_abs_sum(orig)
    idx, x, pos, mask = G7.make_inputs()
    assert abs(wsum - float(z["weight_abs_sum"])) < 1e-9 * wsum
    assert abs(float(x.double().abs().sum()) - float(z["x_abs_sum"])) < 1e-9 * float(z["x_abs_sum"])
    m = qLlamaLayer.QLlamaDecoderLayer(orig, args).to("cuda")
    G.prepare(m, args, {k: v.cuda() for k, v in idx.items()}, quant)
    layers = find_qlinear_layers(m)
    assert len(layers) == 7 and all(l.packed_weight() is not None for l in layers.values())
    rows = torch.from_numpy(z["rows"]).cuda()
    mr = torch.from_numpy(z["mid_rows"]).cuda()
    ref = {k[4:]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("mid_") and k != "mid_rows"}
    xg = x.cuda()

    # ---- (2) stage by stage against the reference's intermediates (16 rows; every stage gets the reference's input)
    differ = lambda a, b: (a != b).float().mean().item()
    fro = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    stage = {}
    xq1 = m.input_layernorm(xg[:, mr])                                   # rows are independent: 16 tokens as a batch
    stage["input_layernorm -> quant"] = differ(xq1[0], ref["xq1"])
    stage["q_proj(input_layernorm)"] = fro(m.self_attn.q_proj(xq1)[0], ref["q"])
    xq2 = m.post_attention_layernorm(ref["h"][None])
    stage["post_attention_layernorm -> quant"] = differ(xq2[0], ref["xq2"])
    act = ops.activate_fp16_i4(ref["gate"], ref["up"], quant_mode="sim", clip=float(args.a_clip_ratio), scale_layout="plain",
                               return_dequant=True)
    stage["silu(gate) * up -> quant"] = differ(act[4], ref["act_q"])
    actq = quant.attach_codes(act[4], quant.ActCodes(act[0], act[1], act[2], act[3], act[4].shape[0], act[4].shape[1]))
    stage["down_proj(quant(silu * up))"] = fro(m.mlp.down_proj(actq), ref["down"])
    gu = m.mlp.gate_proj(xq2)[0], m.mlp.up_proj(xq2)[0]
    stage["gate_proj(post_attention_layernorm)"] = fro(gu[0], ref["gate"])
    stage["up_proj(post_attention_layernorm)"] = fro(gu[1], ref["up"])
    print("stage-wise, teacher-forced against the reference's intermediates:", {k: round(v, 6) for k, v in stage.items()})
    assert stage["input_layernorm -> quant"] <= 5e-3 and stage["post_attention_layernorm -> quant"] <= 5e-3
    assert stage["silu(gate) * up -> quant"] <= 2e-3
    for k in ("q_proj(input_layernorm)", "down_proj(quant(silu * up))", "gate_proj(post_attention_layernorm)", "up_proj(post_attention_layernorm)"):
        assert stage[k] <= 2e-2, (k, stage)

    seen, gemm_fro, codes_hip, codes_ref, mid = {}, {}, {}, {}, {}

    def unpack6(cd):
        """INT4 code fields of an activation operand, whatever its format: F6 records [G, rows_pad, 104] or packed nibbles."""
        if cd.wide == "f6":
            b = cd.o4[:, : cd.rows, :96].reshape(cd.o4.shape[0], cd.rows, 32, 3).to(torch.int32)
            w = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
            f = torch.stack([(w >> (6 * i)) & 0x3F for i in range(4)], -1).reshape(cd.o4.shape[0], cd.rows, 128)
            return torch.where(f == 0x20, torch.zeros_like(f), f)
        a = cd.o4.view(torch.uint8)
        return torch.stack([a & 0xF, a >> 4], -1)

    def hook(name, store, check):
        def f(mod, inp, out):
            cd = quant.get_codes(inp[0])
            assert cd is not None, f"{name}: activation codes lost -> would silently use F.linear"
            store[name] = (unpack6(cd), cd.o8.clone())
            if check:
                assert cd.wide == "f6" and mod._f6 is not None, name        # 2048 rows: the F6 route
                r = torch.nn.functional.linear(inp[0], mod.weight, mod.bias).float()
                # |D_hip - D_ref| <= 1e-2 max(|D_ref|, rms) per element (o_proj's output has entries of ~30 rms, where
                # one fp16 ulp of the reference's own rounding is already 1.5e-2 rms)
                seen[name] = ((out.float() - r).abs() / r.abs().clamp_min(r.pow(2).mean().sqrt())).max().item()
                gemm_fro[name] = ((out.float() - r).norm() / r.norm()).item()
        return f
    handles = [l.register_forward_hook(hook(n_, codes_hip, True)) for n_, l in layers.items()]
    handles.append(m.post_attention_layernorm.register_forward_hook(lambda mod, inp, out: mid.__setitem__("h_hip", inp[0].detach().clone())))
    fused_min = type(m.mlp).FUSED_MIN_ROWS
    m.mlp.FUSED_MIN_ROWS = 1 << 30               # gate_proj / up_proj as their own launches, so that their hooks see them
    y = m(xg, attention_mask=mask.cuda(), position_ids=pos.cuda())[0]
    for h in handles:
        h.remove()
    assert len(seen) == 7 and max(seen.values()) <= 1e-2, seen                          # (1)
    m.mlp.FUSED_MIN_ROWS = fused_min             # and as the module runs them at this size: gate / up / SiLU x up / quantiser fused
    y_fused = m(xg, attention_mask=mask.cuda(), position_ids=pos.cuda())[0]
    assert m.mlp._fused is not None and torch.equal(y_fused, y)                         # same bits (summation order 1 at this shape)
    for l in layers.values():                    # force the reference forward (F.linear on the fake-quant weight)
        l._packed = None
        l._f6 = None
        l._unpackable_key = l._weight_key()
    m.mlp._fused = None
    handles = [l.register_forward_hook(hook(n_, codes_ref, False)) for n_, l in layers.items()]
    handles.append(m.post_attention_layernorm.register_forward_hook(lambda mod, inp, out: mid.__setitem__("h", inp[0].detach().clone())))
    y2 = m(xg, attention_mask=mask.cuda(), position_ids=pos.cuda())[0]
    for h in handles:
        h.remove()
    # the reference-order run again, every GEMM output perturbed by noise of the HIP GEMM's own (measured) relative deviation
    gen = torch.Generator(device="cuda").manual_seed(5)

    def noisy(name):
        def f(mod, inp, out):
            e = torch.randn(out.shape, device=out.device, generator=gen)
            return (out.float() + e * (gemm_fro[name] * out.float().norm() / e.norm())).half()
        return f
    handles = [l.register_forward_hook(noisy(n_)) for n_, l in layers.items()]
    handles.append(m.post_attention_layernorm.register_forward_hook(lambda mod, inp, out: mid.__setitem__("h_noisy", inp[0].detach().clone())))
    y3 = m(xg, attention_mask=mask.cuda(), position_ids=pos.cuda())[0]
    for h in handles:
        h.remove()
    want = torch.from_numpy(z["y_rows"]).cuda()
    hw = torch.from_numpy(z["h_rows"]).cuda()
    rel_h2, rel_h, rel_h3 = fro(mid["h"][0][rows], hw), fro(mid["h_hip"][0][rows], hw), fro(mid["h_noisy"][0][rows], hw)
    rel2, rel, rel3 = fro(y2[0][rows], want), fro(y[0][rows], want), fro(y3[0][rows], want)
    rowsum = y2[0].double().abs().sum(-1)
    rel_rows = ((rowsum - torch.from_numpy(z["y_row_abs_sum"]).cuda().double()).abs() / rowsum).max().item()
    print("HIP GEMM vs F.linear on the same operands, relative Frobenius per projection:", {k: round(v, 6) for k, v in gemm_fro.items()})
    print("vs the reference golden (relative Frobenius, sampled rows) -- attention half: reference-order run", rel_h2, "HIP run", rel_h,
          "perturbed reference-order run", rel_h3, "; block: reference-order run", rel2, "HIP run", rel, "perturbed reference-order run", rel3,
          "; worst row checksum (reference-order run)", rel_rows)
    assert rel_h2 < 1e-2 and rel_h <= 1.25 * rel_h3 + 2e-3 and rel_h < 3e-2                # (3)
    assert rel <= 1.25 * rel3 + 0.01 and rel < 0.25 and rel2 < 0.2
    assert rel_rows < 5e-2 and abs(float(y2.double().abs().sum()) - float(z["y_abs_sum"])) < 5e-3 * float(z["y_abs_sum"])
    flips = {}
    for n_ in layers:
        a4, a8 = codes_hip[n_]
        b4, b8 = codes_ref[n_]
        flips[n_] = ((a4 != b4).float().mean().item(), (a8 != b8).float().mean().item())
    print("code flips per quantiser, HIP run vs reference-order run (INT4, INT8 keeper):", {k: (round(v[0], 5), round(v[1], 5)) for k, v in flips.items()})
    # limits of the 512-wide test, with gate / up at 4 % instead of 2 % (measured 2.8 %: their quantiser sits behind a 2048-token
    # attention and two GEMMs with K = 4096); a GEMM wired to a wrong scale or index flips > 30 % at the next quantiser
    limit = {"self_attn.q_proj": 0.0, "self_attn.k_proj": 0.0, "self_attn.v_proj": 0.0, "self_attn.o_proj": 0.03,
             "mlp.gate_proj": 0.04, "mlp.up_proj": 0.04, "mlp.down_proj": 0.12}
    for n_, (f4, _) in flips.items():
        assert f4 <= limit[n_], (n_, flips)
