"""Golden fixtures for the FLOW-level drop-in tests: the UNMODIFIED reference driver code

    modelutils_llama.reorder_model_llama -> add_act_quant_wrapper_llama -> quantize_model_llama   (modelutils_llama.py:15-153)
    modelutils_llama.quantize_model_gptq_llama (+ the reference's gptq.py solver)                 (modelutils_llama.py:155-266)
    eval.llama_eval                                                                               (eval.py:14-86)

wired as main.py:224-270 wires them, run on CPU with the reference's own classes over the tiny seeded model of tests/flow_model.py.
Build container only (needs /root/reference):

    python tests/golden/gen_golden_flow.py     ->  flow_tokens.npz, flow_rtn_w4a4.npz, flow_gptq_w4a4.npz, flow_rtn_w4a4_gqa.npz

flow_tokens.npz holds the token streams: drawn once, autoregressively, from the UN-quantised tiny model's own distribution (plain
fp32 torch forward below), so that the model is predictive on them and the perplexity of a quantised copy is a sensitive number
instead of ~vocab.  Each (impl, config) runs in a fresh interpreter through tests/flow_run.py.
"""
import math
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


@torch.no_grad()
def fp32_logits(model, ids):
    """Plain fp32 forward of tests/flow_model.TinyLlamaForCausalLM (un-quantised, un-reordered weights)."""
    cfg = model.config
    h = model.model.embed_tokens.weight.float()[ids]
    b, s, d = h.shape
    nh = cfg.num_attention_heads
    hd = d // nh
    pos = torch.arange(s).float()
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    ang = torch.cat([pos[:, None] * inv[None, :]] * 2, -1)
    cos, sin = ang.cos()[None, None], ang.sin()[None, None]
    rot = lambda x: torch.cat((-x[..., hd // 2:], x[..., :hd // 2]), -1)
    mask = torch.full((s, s), float("-inf")).triu(1)
    norm = lambda x, w: x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + cfg.rms_norm_eps) * w.float()
    for L in model.model.layers:
        x = norm(h, L.input_layernorm.weight)
        q, k, v = (torch.nn.functional.linear(x, getattr(L.self_attn, p).weight.float()).view(b, s, -1, hd).transpose(1, 2)
                   for p in ("q_proj", "k_proj", "v_proj"))
        k, v = (t.repeat_interleave(nh // t.shape[1], dim=1) for t in (k, v))          # grouped-query attention: K/V heads shared
        q, k = q * cos + rot(q) * sin, k * cos + rot(k) * sin
        a = torch.softmax(q @ k.transpose(2, 3) / math.sqrt(hd) + mask, -1) @ v
        h = h + torch.nn.functional.linear(a.transpose(1, 2).reshape(b, s, d), L.self_attn.o_proj.weight.float())
        x = norm(h, L.post_attention_layernorm.weight)
        g = torch.nn.functional.silu(torch.nn.functional.linear(x, L.mlp.gate_proj.weight.float()))
        u = torch.nn.functional.linear(x, L.mlp.up_proj.weight.float())
        h = h + torch.nn.functional.linear(g * u, L.mlp.down_proj.weight.float())
    return torch.nn.functional.linear(norm(h, model.model.norm.weight), model.lm_head.weight.float())


@torch.no_grad()
def sample_stream(model, nsamples, seqlen, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, model.config.vocab_size, (nsamples, 1), generator=g)
    while ids.shape[1] < seqlen:
        p = torch.softmax(fp32_logits(model, ids)[:, -1], -1)
        ids = torch.cat([ids, torch.multinomial(p, 1, generator=g)], 1)
    nll = torch.nn.functional.cross_entropy(fp32_logits(model, ids)[:, :-1].reshape(-1, model.config.vocab_size), ids[:, 1:].reshape(-1))
    return ids.reshape(-1), float(nll)


def main():
    from tests.flow_model import TinyLlamaForCausalLM
    from tests.flow_run import CONFIGS
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    tokens = {}
    for name in ("rtn_w4a4", "gptq_w4a4", "rtn_w4a4_gqa"):
        mk, _, seqlens, gptq = CONFIGS[name]
        model = TinyLlamaForCausalLM(**mk).eval()
        for s in seqlens:
            ids, nll = sample_stream(model, 4, s, seed=100 + s)
            tokens[f"{name}.eval_{s}"] = ids.numpy().astype(np.int16)
            print(f"{name}: eval stream seqlen {s}: fp32 model NLL {nll:.3f} (uniform {math.log(mk['vocab']):.3f})")
        if gptq:
            ids, _ = sample_stream(model, 4, seqlens[0], seed=7)
            tokens[f"{name}.calib"] = ids.numpy().astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "flow_tokens.npz"), **tokens)
    for name in ("rtn_w4a4", "gptq_w4a4", "rtn_w4a4_gqa"):
        tok = os.path.join("/tmp", f"flow_tokens_{name}.npz")
        np.savez(tok, **{k.split(".", 1)[1]: v for k, v in tokens.items() if k.startswith(name + ".")})
        out = os.path.join(HERE, f"flow_{name}.npz")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "flow_run.py"), "--impl", "reference", "--config", name,
                            "--tokens", tok, "--out", out], capture_output=True, text=True, cwd=ROOT)
        print(r.stdout[-2000:], r.stderr[-3000:] if r.returncode else "")
        assert r.returncode == 0
        print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
