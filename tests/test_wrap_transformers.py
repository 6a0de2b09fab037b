"""SURVEY 7 hard part 8: the reference's wrappers read attributes that transformers >= 4.45 / 5.x layers no longer have
(`hidden_size`, `num_heads`, `rotary_emb` on the attention module; reference qLlamaLayer.py:170-203).  Our drop-in classes
take a real transformers LlamaDecoderLayer of the installed version.  CPU only: the 16-bit configuration runs no HIP
kernel (model/quant.py returns early for abits >= 16) and must reproduce the wrapped layer itself."""
import types

import torch


def test_wrap_installed_transformers_decoder_layer():
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRotaryEmbedding
    from atom_amd.model import qLlamaLayer
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=512, intermediate_size=1408, num_attention_heads=4, num_key_value_heads=4,
                      num_hidden_layers=1, vocab_size=100, max_position_embeddings=128)
    cfg._attn_implementation = "eager"
    layer = LlamaDecoderLayer(cfg, 0).float().eval()
    args = types.SimpleNamespace(wbits=16, abits=16, a_sym=True, w_sym=True, act_group_size=128, weight_group_size=128,
                                 weight_channel_group=2, keeper=128, keeper_precision=3, a_clip_ratio=0.9, w_clip_ratio=0.85,
                                 kv_clip_ratio=1.0, tiling=0, exponential=False, quant_type="int", static=False,
                                 reorder=False, kv_cache=False)
    x = torch.randn(2, 6, 512)
    pos = torch.arange(6)[None].expand(2, 6)
    cos, sin = LlamaRotaryEmbedding(cfg)(x, pos)
    mask = torch.full((6, 6), float("-inf")).triu(1)[None, None].expand(2, 1, 6, 6)
    with torch.no_grad():
        want = layer(x, attention_mask=mask, position_ids=pos, position_embeddings=(cos, sin))
        want = want[0] if isinstance(want, tuple) else want
        m = qLlamaLayer.QLlamaDecoderLayer(layer, args)
        assert (m.self_attn.num_heads, m.self_attn.head_dim, m.self_attn.hidden_size) == (4, 128, 512)
        got = m(x, attention_mask=mask, position_ids=pos, position_embeddings=(cos, sin))[0]
    assert got.shape == want.shape and torch.allclose(got, want, atol=1e-5, rtol=1e-5)
