"""A tiny seeded Llama-shaped causal LM for the FLOW tests: small enough for the unmodified reference flow to run on
CPU in seconds, shaped so that every object the reference's driver code touches exists:

    model.config.use_cache / .hidden_size, model.seqlen, model.model.embed_tokens / .layers / .norm, model.lm_head, model(batch)
    (modelutils_llama.py:15-266, eval.py:14-86, main.py:199-270)

The decoder layers are instances of the INSTALLED transformers' ``LlamaDecoderLayer`` (the reference dispatches on
``isinstance(layers[i], LlamaDecoderLayer)``) that carry the transformers-4.39 attributes the reference's wrappers read
(qLlamaLayer.py:170-203: hidden_size, num_heads, rotary_emb(x, position_ids), ...), which transformers 5.x no longer has.
Everything is generated from seeds with CPU generators (platform independent); nothing here touches /root/reference.
"""
from __future__ import annotations

import types
from collections import defaultdict

import torch
from torch import nn
from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRMSNorm


class Rotary(nn.Module):
    """transformers-4.39 rotary: forward(x, position_ids) -> (cos, sin) [bsz, seq, head_dim] in x.dtype."""

    def __init__(self, dim, base=10000.0):
        super().__init__()
        self.register_buffer("inv_freq", 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim)), persistent=False)

    @torch.no_grad()
    def forward(self, x, position_ids):
        f = (self.inv_freq[None, :, None].float().to(x.device) @ position_ids[:, None, :].float()).transpose(1, 2)
        emb = torch.cat((f, f), dim=-1)
        return emb.cos().to(x.dtype), emb.sin().to(x.dtype)


def _lin(i, o, s, g):
    l = nn.Linear(i, o, bias=False)
    l.weight.data = torch.randn(o, i, generator=g) * s
    return l.half()


class TinyAttention(nn.Module):
    def __init__(self, cfg, g):
        super().__init__()
        h, nh, kvh = cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads
        self.config = cfg
        self.hidden_size, self.num_heads, self.head_dim = h, nh, h // nh
        self.num_key_value_heads, self.num_key_value_groups = kvh, nh // kvh      # grouped-query attention when kvh < nh (qLlamaLayer.py:268-269)
        self.max_position_embeddings, self.rope_theta = cfg.max_position_embeddings, 10000.0
        self.q_proj, self.k_proj = _lin(h, h, 0.04, g), _lin(h, kvh * (h // nh), 0.04, g)
        self.v_proj, self.o_proj = _lin(h, kvh * (h // nh), 0.04, g), _lin(h, h, 0.04, g)
        self.rotary_emb = Rotary(h // nh)


class TinyMLP(nn.Module):
    def __init__(self, cfg, g):
        super().__init__()
        h, inter = cfg.hidden_size, cfg.intermediate_size
        self.gate_proj, self.up_proj, self.down_proj = _lin(h, inter, 0.04, g), _lin(h, inter, 0.04, g), _lin(inter, h, 0.03, g)
        self.act_fn = nn.SiLU()


class TinyDecoderLayer(LlamaDecoderLayer):
    """isinstance(., LlamaDecoderLayer) holds; the parent constructor (version dependent) is not run."""

    def __init__(self, cfg, g):
        nn.Module.__init__(self)
        self.hidden_size = cfg.hidden_size
        self.self_attn = TinyAttention(cfg, g)
        self.mlp = TinyMLP(cfg, g)
        self.input_layernorm = LlamaRMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps)
        self.post_attention_layernorm = LlamaRMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps)
        for nrm in (self.input_layernorm, self.post_attention_layernorm):
            nrm.weight.data = 1.0 + 0.1 * torch.randn(cfg.hidden_size, generator=g)
        self.input_layernorm.half()
        self.post_attention_layernorm.half()

    def forward(self, *a, **k):          # the flows always replace these layers by QLlamaDecoderLayer before any forward
        raise RuntimeError("TinyDecoderLayer is a container of weights; the flow wraps it in QLlamaDecoderLayer")


class TinyBody(nn.Module):
    def __init__(self, cfg, g):
        super().__init__()
        self.embed_tokens = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        e = torch.randn(cfg.vocab_size, cfg.hidden_size, generator=g)
        e[:, torch.randperm(cfg.hidden_size, generator=g)[:12]] *= 6.0        # outlier channels, as in a trained model
        self.embed_tokens.weight.data = e
        self.embed_tokens.half()
        self.layers = nn.ModuleList([TinyDecoderLayer(cfg, g) for _ in range(cfg.num_hidden_layers)])
        self.norm = LlamaRMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps).half()


class TinyLlamaForCausalLM(nn.Module):
    """model(input_ids [1, T]) -> logits; the layer call passes attention_mask / position_ids as keywords (the Catcher of
    eval.py:26-36 and modelutils_llama.py:172-184 reads exactly these two)."""

    def __init__(self, hidden=512, heads=4, inter=1408, layers=2, vocab=1000, seqlen=96, seed=11, kv_heads=None):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.config = types.SimpleNamespace(hidden_size=hidden, num_attention_heads=heads, num_key_value_heads=kv_heads or heads,
                                            intermediate_size=inter,
                                            num_hidden_layers=layers, vocab_size=vocab, max_position_embeddings=2048,
                                            rms_norm_eps=1e-5, use_cache=False)
        self.seqlen = seqlen
        self.model = TinyBody(self.config, g)
        self.lm_head = _lin(hidden, vocab, 0.12, g)

    @torch.no_grad()
    def forward(self, input_ids):
        h = self.model.embed_tokens(input_ids)
        bsz, seq = input_ids.shape
        pos = torch.arange(seq, device=h.device)[None, :].expand(bsz, seq).contiguous()
        mask = torch.full((seq, seq), torch.finfo(torch.float16).min, device=h.device).triu(1)[None, None]
        mask = mask.expand(bsz, 1, seq, seq).to(h.dtype).contiguous()
        for layer in self.model.layers:
            h = layer(h, attention_mask=mask, position_ids=pos)[0]
        return self.lm_head(self.model.norm(h))


PROJ = (("self_attn", "q_proj"), ("self_attn", "k_proj"), ("self_attn", "v_proj"), ("self_attn", "o_proj"),
        ("mlp", "gate_proj"), ("mlp", "up_proj"), ("mlp", "down_proj"))


def make_reorder_index(model, seed=12):
    """The dictionary get_reorder_index writes (outlier.py:296-340; keys 'layers.{i}.{module}.{proj}.input'): q / k / v share one
    order, gate / up share one.  Seeded random permutations stand in for the sorted activation statistics."""
    g = torch.Generator().manual_seed(seed)
    h, inter = model.config.hidden_size, model.config.intermediate_size
    out = {}
    for i in range(len(model.model.layers)):
        qkv, o, gu, dn = (torch.randperm(h, generator=g), torch.randperm(h, generator=g), torch.randperm(h, generator=g),
                          torch.randperm(inter, generator=g))
        for p in ("q_proj", "k_proj", "v_proj"):
            out[f"layers.{i}.self_attn.{p}.input"] = qkv
        out[f"layers.{i}.self_attn.o_proj.input"] = o
        out[f"layers.{i}.mlp.gate_proj.input"] = gu
        out[f"layers.{i}.mlp.up_proj.input"] = gu
        out[f"layers.{i}.mlp.down_proj.input"] = dn
    return out


def no_scales():
    return defaultdict(lambda: None)          # main.py:240


def paper_args(**over):
    """scripts/run_atom_ppl.sh:11-15 (W4A4 sym, g128, channel_group 2, clips 0.9 / 0.85, keeper 128 INT8, INT4 KV cache)."""
    d = dict(wbits=4, abits=4, a_sym=True, w_sym=True, act_group_size=128, weight_group_size=128, weight_channel_group=2, keeper=128,
             keeper_precision=3, a_clip_ratio=0.9, w_clip_ratio=0.85, kv_clip_ratio=1.0, tiling=0, exponential=False, quant_type="int",
             static=False, reorder=True, kv_cache=True, nsamples=4, percdamp=0.01)
    d.update(over)
    return types.SimpleNamespace(**d)


class TokenStream:
    """what get_loaders returns as `testloader` (datautils.py: a tokenizer output with .input_ids [1, T])"""

    def __init__(self, ids):
        self.input_ids = ids
