"""BASELINE config 4: one Llama-7B decoder block, W4A4 g128 keeper 128, KV fake-quant INT4, batch 32 x seq 2048, through the
drop-in operator surface (atom_amd.model.qLlamaLayer.QLlamaDecoderLayer).  Reports the block time and the time inside the
seven W4A4 GEMMs and the fused quantisers (CUDA events in forward hooks).  Run on the GPU box:  python tools/block_bench.py"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_golden_block as G  # noqa: E402  (pure-torch builders only)
from atom_amd.model import qLlamaLayer, quant  # noqa: E402
from atom_amd.model.qLinearLayer import QLinearLayer  # noqa: E402


WIDE_PROJ = {   # projection -> (fixture key of its input, module path): tests/golden/llama_block_7b_wide.npz
    "q": ("xq1", "self_attn.q_proj"), "k": ("xq1", "self_attn.k_proj"), "v": ("xq1", "self_attn.v_proj"), "o": ("attn_q", "self_attn.o_proj"),
    "gate": ("xq2", "mlp.gate_proj"), "up": ("xq2", "mlp.up_proj"), "down": ("act_q", "mlp.down_proj"),
}


def golden_check(m):
    """the 65,536-token number carries a check.  The timed block IS the model of tests/golden/llama_block_7b_wide.npz
    (same seeds, gen_golden_block7b.py), so its seven W4A4 projections are fed the unmodified REFERENCE's own inputs on the fixture's
    256 token rows and compared with the reference's own outputs on its 256 sampled features (what tests/test_gpu_block.py asserts
    with bounds 1e-3 / 1e-2).  Returns {projection: (relative Frobenius error, worst element / max(|ref|, rms))}."""
    import numpy as np
    from atom_amd import ops
    from atom_amd.model.qLinearLayer import find_qlinear_layers
    z = np.load(os.path.join(ROOT, "tests", "golden", "llama_block_7b_wide.npz"))
    layers = find_qlinear_layers(m)
    out = {}
    for name, (key, path) in WIDE_PROJ.items():
        v = torch.from_numpy(z[key]).cuda().float()
        s4 = torch.from_numpy(z["s4_" + key]).cuda()
        s8 = torch.from_numpy(z["s8_" + key]).cuda()
        M, H = v.shape
        c4 = torch.round(v[:, :-128].reshape(M, -1, 128) / s4.float()[..., None]).reshape(M, H - 128).to(torch.int32)
        c8 = torch.round(v[:, -128:] / s8.float()[:, None]).to(torch.int8)
        o4 = ((c4[:, 0::2] & 0xF) | ((c4[:, 1::2] & 0xF) << 4)).to(torch.uint8).view(torch.int8).contiguous()
        b4, b8, sb, sb8 = layers[path].packed_weight()
        D = ops.dense_layer_gemm_i4_fp16(o4, b4, s4.t().contiguous(), sb, c8.contiguous(), b8, s8, sb8, scale_layout="plain")
        got = D[:, torch.from_numpy(z["cols_" + name]).cuda()].double()
        ref = torch.from_numpy(z["out_" + name]).cuda().double()
        out[name] = (((got - ref).norm() / ref.norm()).item(),
                     ((got - ref).abs() / ref.abs().clamp_min(ref.pow(2).mean().sqrt())).max().item())
    return out


def run(bsz=32, seq=2048, hidden=4096, heads=32, inter=11008, iters=3, warmup=1, verbose=True, attn_sdpa=False, check=False, keep_f6=True):
    """Returns dict(block_ms, gemm_ms, gemm_tops, spans={module: ms}) -- also what `bench.py --workload block` reports.
    attn_sdpa: the configuration opt-in args.attn_sdpa (torch's fused attention instead of the reference's materialised score matrix).
    check (Llama-7B width only): + "golden" = golden_check() of the very model that was timed."""
    args = types.SimpleNamespace(wbits=4, abits=4, a_sym=True, w_sym=True, act_group_size=128, weight_group_size=128,
                                 weight_channel_group=2, keeper=128, keeper_precision=3, a_clip_ratio=0.9,
                                 w_clip_ratio=0.85, kv_clip_ratio=1.0, tiling=0, exponential=False, quant_type="int",
                                 static=False, reorder=True, kv_cache=True, attn_sdpa=bool(attn_sdpa))
    if (hidden, heads, inter) == (4096, 32, 11008):           # the model of the 7B-width goldens (weights seed 17, reorder indices seed 18)
        import gen_golden_block7b as G7
        orig = G7.build_original()
        idx = G.make_inputs(hidden, inter, 1, 8, seed=G7.SEED_X)[0]
    else:
        check = False
        orig = G.build_original(hidden, heads, inter, seed=7)
        idx, _, _, _ = G.make_inputs(hidden, inter, 1, 8, seed=8)
    m = qLlamaLayer.QLlamaDecoderLayer(orig, args).to("cuda")
    G.prepare(m, args, {k: v.cuda() for k, v in idx.items()}, quant)
    if not keep_f6:                                           # 4-bit weights only in HBM: the BF6 form re-made by every call (QLinearLayer.keep_f6)
        for mod in m.modules():
            if type(mod) is QLinearLayer:
                mod.keep_f6 = False
    x = torch.randn(bsz, seq, hidden, device="cuda").half()
    pos = torch.arange(seq, device="cuda")[None, :].expand(bsz, seq).contiguous()
    mask = torch.full((seq, seq), torch.finfo(torch.float16).min, device="cuda").triu(1)[None, None].expand(bsz, 1, seq, seq).half().contiguous()

    spans = {}

    def timed(name, mod):
        def pre(_m, _i):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            spans.setdefault(name, []).append([e, None])

        def post(_m, _i, _o):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            spans[name][-1][1] = e
        mod.register_forward_pre_hook(pre)
        mod.register_forward_hook(post)

    # the MLP is timed as a whole: from 512 tokens gate_proj / up_proj / SiLU x up / quantiser are ONE launch (no per-layer hooks fire)
    for name, mod in m.named_modules():
        if (type(mod) is QLinearLayer and not name.startswith("mlp.")) or name in ("input_layernorm", "post_attention_layernorm", "mlp"):
            timed(name, mod)
    with torch.no_grad():
        for _ in range(max(warmup, 1)):
            m(x, attention_mask=mask, position_ids=pos)        # warm-up (packs nothing: quant() already packed)
        torch.cuda.synchronize()
        spans.clear()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            m(x, attention_mask=mask, position_ids=pos)
        e1.record()
        torch.cuda.synchronize()
    total = e0.elapsed_time(e1) / iters
    if verbose:
        print(f"BLOCK batch={bsz} seq={seq} hidden={hidden} inter={inter}: {total:.2f} ms per block forward", flush=True)
    gemm = 0.0
    per = {}
    for name, evs in spans.items():
        ms = sum(a.elapsed_time(b) for a, b in evs) / iters
        per[name] = round(ms, 4)
        if verbose:
            print(f"  {name:28s} {ms:8.3f} ms")
        if "proj" in name or name == "mlp":
            gemm += ms
    M = bsz * seq
    ops = 2.0 * M * (4 * hidden * hidden + 3 * hidden * inter)
    if verbose:
        print(f"  seven W4A4 GEMMs (the MLP's three incl. its fused SiLU x up quantiser): {gemm:.2f} ms = {ops / gemm / 1e9:.0f} TOPS; "
              f"rest (attention in torch, KV fake-quant, RoPE, residuals): {total - gemm:.2f} ms")
    res = {"block_ms": total, "gemm_ms": gemm, "gemm_ops": ops, "gemm_tops": ops / gemm / 1e9, "spans": per, "tokens": M}
    if check:
        with torch.no_grad():
            res["golden"] = golden_check(m)
        if verbose:
            print("  the timed model's projections on the reference's inputs vs the reference's outputs (256 rows x 256 features: relative "
                  "Frobenius, worst element): " + ", ".join(f"{k} {a:.1e} {b:.1e}" for k, (a, b) in res["golden"].items()))
    return res


if __name__ == "__main__":
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    run(bsz=b, check=True, keep_f6=not (len(sys.argv) > 2 and sys.argv[2] == "nibble"))
