"""Golden vectors for ONE Llama-7B-width decoder block from the UNMODIFIED reference (BASELINE config 4 at its stated width, at the
batch the CPU oracle can hold -- SURVEY 8(d): "CPU oracle at reduced batch (B=1) because the naive attention materialises
[B,32,2048,2048]"): hidden 4096 = 32 heads x 128, intermediate 11008, batch 1 x seq 2048, KV-cache INT4 fake-quant, W4A4 g128
keeper 128, clips 0.9 / 0.85.

Runs reference model/qLlamaLayer.py:86-127 (QLlamaDecoderLayer.forward) on CPU through the duck-typed "original layer" of
gen_golden_block.py, prepared the way modelutils_llama.py prepares a layer (reorder -> quantiser wrappers -> quant()).  The output
is 2048 x 4096 halves (16 MB); the fixture keeps a FIXED SAMPLE of token rows (every 32nd row + the first and last 8 = 80 rows,
0.6 MB compressed) of y and of the residual-stream input of the MLP half, 16 rows of the intermediate tensors either side of every
quantiser (for stage-wise, teacher-forced checks), checksums of the full tensors, and checksums of the weights and of x, which the test re-generates from the same seeds (torch CPU generators are platform-independent).
A second file, llama_block_7b_wide.npz, is the WIDE teacher-forcing sample: for 256 token rows (every 8th) the reference's input of
every projection -- the four fake-quantised activation tensors, which deflate to a third of their size -- and 256 sampled output
features of all seven projections, so that each GEMM is checked against the reference's own output on 256 x 256 elements per
projection with the reference's own input (a 5 % systematic error in one projection is 100 sigma there).
Container only (needs /root/reference; ~2 min on 8 cores):

    python tests/golden/gen_golden_block7b.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import _import_reference, paper_args, n  # noqa: E402
import gen_golden_block as GB  # noqa: E402

HIDDEN, HEADS, INTER, BSZ, SEQ = 4096, 32, 11008, 1, 2048
SEED_W, SEED_X = 17, 18


def sample_rows(seq=SEQ):
    """The token rows the fixture stores: every 32nd row plus the first and the last 8."""
    rows = sorted(set(range(0, seq, 32)) | set(range(8)) | set(range(seq - 8, seq)))
    return np.array(rows, dtype=np.int64)


def mid_rows(seq=SEQ):
    """The 16 token rows for which the fixture also stores intermediate tensors (stage-wise, teacher-forced checks)."""
    return np.array(sorted(set(range(5, seq, 136)) | {0})[:16], dtype=np.int64)


def wide_rows(seq=SEQ):
    """The 256 token rows of the wide teacher-forcing sample."""
    return np.arange(0, seq, seq // 256, dtype=np.int64)[:256]


def wide_cols(nfeat):
    """256 output features of a projection: a fixed pseudo-random subset (every 128-feature tile of the GEMM kernels is hit)."""
    g = np.random.default_rng(1234 + nfeat)
    return np.sort(g.choice(nfeat, 256, replace=False)).astype(np.int64)


def recover_scales(v, qmax, group):
    """The per-(row, group) fp16 scale s of a fake-quantised tensor v = half(code * s) (model/quant.py:141-181 returns only v):
    the largest-code candidate s = half(max|v| / k), k = qmax .. 1 (and its fp16 neighbours), for which every element is
    reproduced EXACTLY by an integer code in [-qmax - 1, qmax].  Where several (code, s) pairs describe the same numbers (all codes
    even) they differ by a power of two and the W4A4 GEMM contract gives identical sums.  Returns s [rows, groups] fp16; raises if
    a group cannot be reproduced."""
    r, h = v.shape
    x = v.astype(np.float32).reshape(r, h // group, group)
    vmax = np.abs(x).max(-1)
    best = np.zeros(vmax.shape, dtype=np.float16)
    done = vmax == 0
    best[done] = 1.0
    for k in range(qmax + 1, 0, -1):
        base = (vmax / k).astype(np.float16)
        for d in (0, 1, -1):
            cand = (base.view(np.int16) + d).view(np.float16)
            cf = cand.astype(np.float32)
            with np.errstate(divide="ignore", invalid="ignore"):
                c = np.rint(x / cf[..., None])
            ok = (np.abs(c).max(-1) <= qmax + 1) & (c.max(-1) <= qmax) & np.isfinite(cf) & (cf > 0)
            ok &= ((c * cf[..., None]).astype(np.float16) == x.astype(np.float16)).all(-1)
            take = ok & ~done
            best[take] = cand[take]
            done |= take
    assert done.all(), "fake-quantised group not reproducible by integer codes: %d of %d" % ((~done).sum(), done.size)
    return best


def build_original():
    """Llama-7B-width duck-typed layer: weights ~ N(0, 0.02) (HF initializer_range), norm weights 1 + 0.1 N(0, 1)."""
    from transformers.models.llama.modeling_llama import LlamaRMSNorm
    import types
    g = torch.Generator().manual_seed(SEED_W)
    lin = lambda i, o, s: GB._lin(i, o, s, g)
    attn = types.SimpleNamespace(config=None, hidden_size=HIDDEN, num_heads=HEADS, num_key_value_heads=HEADS,
                                 num_key_value_groups=1, max_position_embeddings=4096, rope_theta=10000.0,
                                 q_proj=lin(HIDDEN, HIDDEN, 0.02), k_proj=lin(HIDDEN, HIDDEN, 0.02),
                                 v_proj=lin(HIDDEN, HIDDEN, 0.02), o_proj=lin(HIDDEN, HIDDEN, 0.02),
                                 rotary_emb=GB.Rotary(HIDDEN // HEADS))
    mlp = types.SimpleNamespace(gate_proj=lin(HIDDEN, INTER, 0.02), up_proj=lin(HIDDEN, INTER, 0.02),
                                down_proj=lin(INTER, HIDDEN, 0.02), act_fn=torch.nn.SiLU())
    n1, n2 = LlamaRMSNorm(HIDDEN, eps=1e-5), LlamaRMSNorm(HIDDEN, eps=1e-5)
    n1.weight.data = 1.0 + 0.1 * torch.randn(HIDDEN, generator=g)
    n2.weight.data = 1.0 + 0.1 * torch.randn(HIDDEN, generator=g)
    return types.SimpleNamespace(hidden_size=HIDDEN, self_attn=attn, mlp=mlp, input_layernorm=n1.half(),
                                 post_attention_layernorm=n2.half())


def weight_abs_sum(orig):
    s = 0.0
    for mod in (orig.self_attn.q_proj, orig.self_attn.k_proj, orig.self_attn.v_proj, orig.self_attn.o_proj,
                orig.mlp.gate_proj, orig.mlp.up_proj, orig.mlp.down_proj):
        s += float(mod.weight.detach().double().abs().sum())
    return s + float(orig.input_layernorm.weight.detach().double().abs().sum()
                     + orig.post_attention_layernorm.weight.detach().double().abs().sum())


def make_inputs():
    """Reorder indices, x ~ N(0, 1) [1, 2048, 4096] with 32 outlier channels x 8, positions, causal mask."""
    return GB.make_inputs(HIDDEN, INTER, BSZ, SEQ, seed=SEED_X)


def main():
    quant, qLinearLayer, qLlamaLayer = _import_reference()
    torch.set_num_threads(os.cpu_count() or 8)
    args = paper_args(kv_cache=True)
    t0 = time.time()
    orig = build_original()
    wsum = weight_abs_sum(orig)
    idx, x, pos, mask = make_inputs()
    m = qLlamaLayer.QLlamaDecoderLayer(orig, args)
    GB.prepare(m, args, idx, quant)
    print("prepared in %.1f s" % (time.time() - t0))
    mid = {}
    keep = lambda name, io: (lambda mod, inp, out: mid.__setitem__(name, (inp[0] if io == "in" else out).detach().clone()))
    hooks = [m.post_attention_layernorm.register_forward_hook(keep("h", "in")),
             m.input_layernorm.register_forward_hook(keep("xq1", "out")),
             m.post_attention_layernorm.register_forward_hook(keep("xq2", "out")),
             m.self_attn.q_proj.register_forward_hook(keep("q", "out")),
             m.self_attn.o_proj.register_forward_hook(keep("attn_q", "in")),
             m.mlp.gate_proj.register_forward_hook(keep("gate", "out")),
             m.mlp.up_proj.register_forward_hook(keep("up", "out")),
             m.mlp.down_proj.register_forward_hook(keep("act_q", "in")),
             m.mlp.down_proj.register_forward_hook(keep("down", "out"))]
    wide = {}
    keepw = lambda name, io: (lambda mod, inp, out: wide.__setitem__(name, (inp[0] if io == "in" else out).detach().clone()))
    hooks += [m.self_attn.k_proj.register_forward_hook(keepw("k", "out")), m.self_attn.v_proj.register_forward_hook(keepw("v", "out")),
              m.self_attn.o_proj.register_forward_hook(keepw("o", "out"))]
    t0 = time.time()
    with torch.no_grad():
        y = m(x.clone(), attention_mask=mask, position_ids=pos)[0]
    for h in hooks:
        h.remove()
    print("reference QLlamaDecoderLayer.forward at 1 x %d x %d on CPU: %.1f s" % (SEQ, HIDDEN, time.time() - t0))
    rows = sample_rows()
    yf = y[0].float()
    mr = mid_rows()
    out = dict(rows=rows, y_rows=n(y[0][rows]), h_rows=n(mid["h"][0][rows]), mid_rows=mr,
               **{"mid_" + k: n(v[0][mr]) for k, v in mid.items()},
               y_abs_sum=np.float64(yf.double().abs().sum().item()), y_sq_sum=np.float64(yf.double().pow(2).sum().item()),
               y_row_abs_sum=n(yf.double().abs().sum(-1)).astype(np.float32),       # one number per token row: localises a mismatch
               x_abs_sum=np.float64(x.double().abs().sum().item()), weight_abs_sum=np.float64(wsum))
    path = os.path.join(HERE, "llama_block_7b_1x2048.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; |y| rms", float(yf.pow(2).mean().sqrt()))
    wr = wide_rows()
    outs = dict(q=mid["q"], k=wide["k"], v=wide["v"], o=wide["o"], gate=mid["gate"], up=mid["up"], down=mid["down"])
    w = dict(rows=wr, xq1=n(mid["xq1"][0][wr]), attn_q=n(mid["attn_q"][0][wr]), xq2=n(mid["xq2"][0][wr]), act_q=n(mid["act_q"][0][wr]))
    for k in ("xq1", "attn_q", "xq2", "act_q"):                  # the integer view of the reference's tensors: scales per (row, group)
        w["s4_" + k] = recover_scales(w[k][:, :-128], 7, 128)
        w["s8_" + k] = recover_scales(w[k][:, -128:], 127, 128)[:, 0]
    for k, v in outs.items():
        cols = wide_cols(v.shape[-1])
        w["cols_" + k] = cols
        w["out_" + k] = n(v[0][wr][:, cols])
    path = os.path.join(HERE, "llama_block_7b_wide.npz")
    np.savez_compressed(path, **w)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
