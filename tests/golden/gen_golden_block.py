"""Golden vectors for a (reduced) Llama decoder block from the UNMODIFIED reference (BASELINE config 4 shape family:
hidden 512 = 4 heads x 128, intermediate 1408, seq 24, batch 2, KV-cache INT4 fake-quant, W4A4 g128 keeper 128).

Runs reference model/qLlamaLayer.py on CPU through a duck-typed "original layer" (transformers 5.x no longer exposes the
attributes the reference reads, SURVEY 7 hard part 8).  Container only:  python tests/golden/gen_golden_block.py
"""
import os
import sys
import types
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import _import_reference, paper_args, n  # noqa: E402


class Rotary(torch.nn.Module):
    """HF-4.39-style rotary: forward(x, position_ids) -> (cos, sin) [bsz, seq, head_dim] in x.dtype."""

    def __init__(self, dim, base=10000.0):
        super().__init__()
        self.register_buffer("inv_freq", 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim)), persistent=False)

    @torch.no_grad()
    def forward(self, x, position_ids):
        f = (self.inv_freq[None, :, None].float() @ position_ids[:, None, :].float()).transpose(1, 2)
        emb = torch.cat((f, f), dim=-1)
        return emb.cos().to(x.dtype), emb.sin().to(x.dtype)


def build_original(hidden, heads, inter, seed):
    from transformers.models.llama.modeling_llama import LlamaRMSNorm
    g = torch.Generator().manual_seed(seed)
    lin = lambda i, o, s: _lin(i, o, s, g)
    attn = types.SimpleNamespace(config=None, hidden_size=hidden, num_heads=heads, num_key_value_heads=heads,
                                 num_key_value_groups=1, max_position_embeddings=2048, rope_theta=10000.0,
                                 q_proj=lin(hidden, hidden, 0.04), k_proj=lin(hidden, hidden, 0.04),
                                 v_proj=lin(hidden, hidden, 0.04), o_proj=lin(hidden, hidden, 0.04),
                                 rotary_emb=Rotary(hidden // heads))
    mlp = types.SimpleNamespace(gate_proj=lin(hidden, inter, 0.04), up_proj=lin(hidden, inter, 0.04),
                                down_proj=lin(inter, hidden, 0.03), act_fn=torch.nn.SiLU())
    n1, n2 = LlamaRMSNorm(hidden, eps=1e-5), LlamaRMSNorm(hidden, eps=1e-5)
    n1.weight.data = 1.0 + 0.1 * torch.randn(hidden, generator=g)
    n2.weight.data = 1.0 + 0.1 * torch.randn(hidden, generator=g)
    return types.SimpleNamespace(hidden_size=hidden, self_attn=attn, mlp=mlp, input_layernorm=n1.half(),
                                 post_attention_layernorm=n2.half())


def _lin(i, o, s, g):
    l = torch.nn.Linear(i, o, bias=False)
    l.weight.data = torch.randn(o, i, generator=g) * s
    return l.half()


def prepare(m, args, idx, quant_mod):
    """The steps of modelutils_llama.py: reorder_model_llama (:15-75), add_act_quant_wrapper_llama (:77-124),
    quantize_model_llama (:126-153), on one layer."""
    m.mlp.gate_proj.reorder(in_reorder_index=idx["gateup"], out_reorder_index=idx["down"])
    m.mlp.up_proj.reorder(in_reorder_index=idx["gateup"], out_reorder_index=idx["down"])
    m.mlp.down_proj.reorder(in_reorder_index=idx["down"], out_reorder_index=None)
    for p in ("q_proj", "k_proj", "v_proj"):
        getattr(m.self_attn, p).reorder(in_reorder_index=idx["qkv"], out_reorder_index=None)
    m.self_attn.o_proj.reorder(in_reorder_index=idx["o"], out_reorder_index=None)
    m.input_layernorm.register_buffer("reorder_index", idx["qkv"])
    m.post_attention_layernorm.register_buffer("reorder_index", idx["gateup"])
    m.self_attn.register_buffer("reorder_index", idx["o"])
    act = partial(quant_mod.quantize_activation_wrapper, args=args)
    m.self_attn.act_quant.configure(act, None)
    m.self_attn.v_quant.configure(partial(quant_mod.quantize_attn_v_wrapper, args=args), None)
    m.self_attn.k_quant.configure(partial(quant_mod.quantize_attn_k_wrapper, args=args), None)
    m.mlp.act_quant.configure(act, None)
    m.input_layernorm.act_quant.configure(act, None)
    m.post_attention_layernorm.act_quant.configure(act, None)
    for lin in (m.mlp.gate_proj, m.mlp.up_proj, m.mlp.down_proj, m.self_attn.q_proj, m.self_attn.k_proj,
                m.self_attn.v_proj, m.self_attn.o_proj):
        lin.quant()


def make_inputs(hidden, inter, bsz, seq, seed):
    g = torch.Generator().manual_seed(seed)
    idx = {k: torch.randperm(hidden, generator=g) for k in ("qkv", "o", "gateup")}
    idx["down"] = torch.randperm(inter, generator=g)
    x = torch.randn(bsz, seq, hidden, generator=g).half()
    x[..., torch.randperm(hidden, generator=g)[:16]] *= 8.0
    pos = torch.arange(seq)[None, :].expand(bsz, seq).contiguous()
    mask = torch.full((seq, seq), torch.finfo(torch.float16).min).triu(1)[None, None].expand(bsz, 1, seq, seq).half()
    return idx, x, pos, mask.contiguous()


def main():
    quant, qLinearLayer, qLlamaLayer = _import_reference()
    torch.set_num_threads(4)
    hidden, heads, inter, bsz, seq = 512, 4, 1408, 2, 24
    args = paper_args(kv_cache=True)
    orig = build_original(hidden, heads, inter, seed=7)
    weights = {f"attn.{k}": n(getattr(orig.self_attn, k).weight) for k in ("q_proj", "k_proj", "v_proj", "o_proj")}
    weights.update({f"mlp.{k}": n(getattr(orig.mlp, k).weight) for k in ("gate_proj", "up_proj", "down_proj")})
    weights["norm1"] = n(orig.input_layernorm.weight)
    weights["norm2"] = n(orig.post_attention_layernorm.weight)
    idx, x, pos, mask = make_inputs(hidden, inter, bsz, seq, seed=8)
    m = qLlamaLayer.QLlamaDecoderLayer(orig, args)
    prepare(m, args, idx, quant)
    with torch.no_grad():
        y = m(x.clone(), attention_mask=mask, position_ids=pos)[0]
    # weights / indices / inputs are re-generated from the seeds by the test (torch CPU generators are
    # platform-independent); only the reference OUTPUT and a checksum of the inputs are stored.
    csum = float(sum(np.abs(v.astype(np.float64)).sum() for v in weights.values()))
    out = dict(x=n(x), y=n(y), weight_abs_sum=np.float64(csum))
    path = os.path.join(HERE, "llama_block_512.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; |y| rms", float(y.float().pow(2).mean().sqrt()))


if __name__ == "__main__":
    main()
