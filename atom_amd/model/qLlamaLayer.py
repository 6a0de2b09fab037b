"""HIP-backed mirror of the reference's ``model/qLlamaLayer.py``: QLlamaDecoderLayer / QLlamaRMSNorm / QLlamaAttention /
QLlamaMLP with the same constructor and forward signatures (reference file:line in each docstring).

What changes underneath, for the paper configuration on a GPU:
  * QLlamaRMSNorm.forward   = ONE HIP kernel: RMSNorm * w -> channel gather -> per-token INT4/INT8 quant
                              (atom_rmsnorm_reorder_quant_f16) instead of norm + index_select + 6 elementwise passes;
  * QLlamaMLP.forward       = gate/up W4A4 GEMMs -> ONE HIP kernel silu(gate)*up -> quant -> down W4A4 GEMM;
  * QLlamaAttention.forward = q/k/v W4A4 GEMMs; attention itself in torch (out of the hot-path scope);
                              gather + quant of the context in one HIP kernel -> o_proj W4A4 GEMM.
The wrappers only READ attributes of the wrapped HF modules (duck-typed), so they work with transformers 4.39 (the
reference's pin) and 5.x module layouts alike.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import nn

from .. import ops as _ops
from .qLinearLayer import QLinearLayer
from .quant import (ActCodes, Quantizer, _reorder_index_i16, attach_codes, get_codes, hip_act_quant,  # noqa: F401
                    want_wide_codes)


def rotate_half(x):
    half = x.shape[-1] // 2
    return torch.cat((-x[..., half:], x[..., :half]), dim=-1)


def apply_rotary_pos_emb(q, k, cos, sin, position_ids=None, unsqueeze_dim=1):
    cos, sin = cos.unsqueeze(unsqueeze_dim), sin.unsqueeze(unsqueeze_dim)
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


def repeat_kv(hidden_states: torch.Tensor, n_rep: int) -> torch.Tensor:
    if n_rep == 1:
        return hidden_states
    b, h, s, d = hidden_states.shape
    return hidden_states[:, :, None].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


def _cfg(obj, name, default=None):
    """Attribute of an HF module, falling back to its config (4.39 has module attributes, 5.x only the config)."""
    if hasattr(obj, name):
        return getattr(obj, name)
    cfg = getattr(obj, "config", None)
    alias = {"num_heads": "num_attention_heads"}.get(name, name)
    if cfg is not None and hasattr(cfg, alias):
        return getattr(cfg, alias)
    return default


class QLlamaDecoderLayer(nn.Module):
    """reference qLlamaLayer.py:52-127."""

    def __init__(self, originalLayer, args):
        super().__init__()
        self.args = args
        self.hidden_size = _cfg(originalLayer, "hidden_size", _cfg(originalLayer.self_attn, "hidden_size"))
        self.self_attn = QLlamaAttention(originalLayer.self_attn, args)
        self.mlp = QLlamaMLP(originalLayer.mlp, args)
        self.input_layernorm = QLlamaRMSNorm(originalLayer.input_layernorm, args)
        self.post_attention_layernorm = QLlamaRMSNorm(originalLayer.post_attention_layernorm, args)

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        self.self_attn = self.self_attn.to(*args, **kwargs)
        self.input_layernorm = self.input_layernorm.to(*args, **kwargs)
        self.post_attention_layernorm = self.post_attention_layernorm.to(*args, **kwargs)
        self.mlp = self.mlp.to(*args, **kwargs)
        return self

    @torch.no_grad()
    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                output_attentions: Optional[bool] = False, use_cache: Optional[bool] = False, **kwargs):
        residual = hidden_states
        hidden_states = self.input_layernorm(hidden_states)
        hidden_states, self_attn_weights, present_key_value = self.self_attn(
            hidden_states=hidden_states, attention_mask=attention_mask, position_ids=position_ids,
            past_key_value=past_key_value, output_attentions=output_attentions, use_cache=use_cache, **kwargs)
        hidden_states = residual + hidden_states
        residual = hidden_states
        hidden_states = self.post_attention_layernorm(hidden_states)
        hidden_states = residual + self.mlp(hidden_states)
        outputs = (hidden_states,)
        if output_attentions:
            outputs += (self_attn_weights,)
        if use_cache:
            outputs += (present_key_value,)
        return outputs


class QLlamaRMSNorm(nn.Module):
    """reference qLlamaLayer.py:129-158: originalNorm -> index_select(reorder_index) -> act_quant."""

    def __init__(self, originalNorm, args):
        super().__init__()
        self.originalNorm = originalNorm
        self.act_quant = Quantizer(args=args)
        self.register_buffer("reorder_index", None)
        self.args = args

    @torch.no_grad()
    def forward(self, hidden_states):
        hidden = hidden_states.shape[-1]
        hot = self.act_quant.hot_args(hidden) if self.args.abits < 16 else None
        w = getattr(self.originalNorm, "weight", None)
        eps = getattr(self.originalNorm, "variance_epsilon", getattr(self.originalNorm, "eps", None))
        if (hot is not None and hidden_states.is_cuda and hidden_states.dtype == torch.float16 and w is not None
                and eps is not None and w.dtype == torch.float16):
            shape = hidden_states.shape
            x2 = hidden_states.reshape(-1, hidden)
            if not x2.is_contiguous():
                x2 = x2.contiguous()
            idx = self.reorder_index
            if idx is None:
                idx = torch.arange(hidden, device=x2.device)
            wide = want_wide_codes(x2.shape[0])
            o8, o4, s8, s4, xq = _ops.rmsnorm_fp16_i4(x2, w, _reorder_index_i16(idx, x2.device), float(eps),
                                                      quant_mode="sim", clip=float(hot.a_clip_ratio),
                                                      scale_layout="plain", return_dequant=True, wide_codes=wide)
            return attach_codes(xq.view(shape), ActCodes(o8, o4, s8, s4, x2.shape[0], hidden, wide=wide))
        result = self.originalNorm(hidden_states)
        if self.reorder_index is not None:
            assert result.shape[-1] == self.reorder_index.shape[0]
            result = torch.index_select(result, result.dim() - 1, self.reorder_index)
        if self.args.abits < 16:
            result = self.act_quant(result)
        return result

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        self.originalNorm = self.originalNorm.to(*args, **kwargs)
        self.act_quant = self.act_quant.to(*args, **kwargs)
        if self.reorder_index is not None:
            self.reorder_index = self.reorder_index.to(*args, **kwargs)
        return self


class QLlamaAttention(nn.Module):
    """reference qLlamaLayer.py:160-311."""

    def __init__(self, originalAttn, args):
        super().__init__()
        self.abits = args.abits
        self.q_kv_cache = args.kv_cache
        # opt-in through the configuration namespace the reference passes around (args.attn_sdpa = True; absent = False): fused attention
        # instead of the reference's materialised score matrix.  (Rounds 2-4 read an environment variable at import.)
        self.use_sdpa = bool(getattr(args, "attn_sdpa", False))
        self.config = getattr(originalAttn, "config", None)
        self.hidden_size = _cfg(originalAttn, "hidden_size")
        self.num_heads = _cfg(originalAttn, "num_heads")
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = _cfg(originalAttn, "num_key_value_heads", self.num_heads)
        self.num_key_value_groups = _cfg(originalAttn, "num_key_value_groups", self.num_heads // self.num_key_value_heads)
        self.max_position_embeddings = _cfg(originalAttn, "max_position_embeddings")
        self.rope_theta = _cfg(originalAttn, "rope_theta")
        if self.head_dim * self.num_heads != self.hidden_size:
            raise ValueError(f"hidden_size {self.hidden_size} is not a multiple of num_heads {self.num_heads}")
        self.q_proj = QLinearLayer(originalAttn.q_proj, args)
        self.k_proj = QLinearLayer(originalAttn.k_proj, args)
        self.v_proj = QLinearLayer(originalAttn.v_proj, args)
        self.o_proj = QLinearLayer(originalAttn.o_proj, args)
        self.rotary_emb = getattr(originalAttn, "rotary_emb", None)
        self.act_quant = Quantizer(args=args)
        self.v_quant = Quantizer(args=args)
        self.k_quant = Quantizer(args=args)
        self.register_buffer("reorder_index", None)

    def _shape(self, tensor: torch.Tensor, seq_len: int, bsz: int):
        return tensor.view(bsz, seq_len, self.num_heads, self.head_dim).transpose(1, 2).contiguous()

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        for name in ("q_proj", "k_proj", "v_proj", "o_proj", "act_quant", "v_quant", "k_quant"):
            setattr(self, name, getattr(self, name).to(*args, **kwargs))
        if self.rotary_emb is not None:
            self.rotary_emb = self.rotary_emb.to(*args, **kwargs)
        if self.reorder_index is not None:
            self.reorder_index = self.reorder_index.to(*args, **kwargs)
        return self

    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.LongTensor] = None, past_key_value: Optional[Tuple[torch.Tensor]] = None,
                output_attentions: bool = False, use_cache: bool = False, position_embeddings=None, **kwargs):
        bsz, q_len, _ = hidden_states.size()
        # three W4A4 GEMMs on the same activation codes
        q = self.q_proj(hidden_states).view(bsz, q_len, self.num_heads, self.head_dim).transpose(1, 2)
        k = self.k_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        v = self.v_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        kv_seq_len = k.shape[-2] + (past_key_value[0].shape[-2] if past_key_value is not None else 0)

        if self.q_kv_cache:
            k = self.k_quant(k)                                   # before RoPE (reference :248-249)
        if position_embeddings is not None:                       # transformers >= 4.45 passes (cos, sin) down
            cos, sin = position_embeddings
        else:
            cos, sin = self.rotary_emb(v, position_ids)
        q, k = apply_rotary_pos_emb(q, k, cos, sin)

        if past_key_value is not None:
            k = torch.cat([past_key_value[0], k], dim=2)
            v = torch.cat([past_key_value[1], v], dim=2)
        past_key_value = (k, v) if use_cache else None
        k = repeat_kv(k, self.num_key_value_groups)
        v = repeat_kv(v, self.num_key_value_groups)

        if self.q_kv_cache:
            v = self.v_quant(v)
        if self.use_sdpa and not output_attentions:
            # opt-in (args.attn_sdpa): fused attention instead of the reference's materialised score matrix
            # (qLlamaLayer.py:262-290).  Same mathematics, FP32 softmax inside the kernel; not bit-identical.
            attn_output = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask)
            attn_weights = None
            return self._finish(attn_output, bsz, q_len, attn_weights, past_key_value, output_attentions)
        attn_weights = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(self.head_dim)
        if tuple(attn_weights.shape) != (bsz, self.num_heads, q_len, kv_seq_len):
            raise ValueError(f"score tensor has shape {tuple(attn_weights.shape)}, expected "
                             f"{(bsz, self.num_heads, q_len, kv_seq_len)}")
        if attention_mask is not None:
            if tuple(attention_mask.shape) != (bsz, 1, q_len, kv_seq_len):
                raise ValueError(f"mask has shape {tuple(attention_mask.shape)}, expected {(bsz, 1, q_len, kv_seq_len)}")
            attn_weights = attn_weights + attention_mask
        attn_weights = nn.functional.softmax(attn_weights, dim=-1, dtype=torch.float32).to(q.dtype)
        attn_output = torch.matmul(attn_weights, v)
        return self._finish(attn_output, bsz, q_len, attn_weights, past_key_value, output_attentions)

    def _finish(self, attn_output, bsz, q_len, attn_weights, past_key_value, output_attentions):
        assert tuple(attn_output.shape) == (bsz, self.num_heads, q_len, self.head_dim)
        attn_output = attn_output.transpose(1, 2).contiguous().reshape(bsz, q_len, self.hidden_size)

        # gather (reorder_index) + quantise the context for o_proj: one HIP kernel in the hot configuration
        hot = self.act_quant.hot_args(self.hidden_size)
        if hot is not None and attn_output.is_cuda and attn_output.dtype == torch.float16:
            attn_output = hip_act_quant(attn_output, hot, self.reorder_index)
        else:
            if self.reorder_index is not None:
                attn_output = torch.index_select(attn_output, 2, self.reorder_index)
            attn_output = self.act_quant(attn_output)
        attn_output = self.o_proj(attn_output)
        if not output_attentions:
            attn_weights = None
        return attn_output, attn_weights, past_key_value


class QLlamaMLP(nn.Module):
    """reference qLlamaLayer.py:314-351.  No runtime gather: gate/up rows were pre-permuted with down_proj's input
    order (modelutils_llama.py:33-40)."""

    def __init__(self, originalMLP, args):
        super().__init__()
        self.gate_proj = QLinearLayer(originalMLP.gate_proj, args)
        self.down_proj = QLinearLayer(originalMLP.down_proj, args)
        self.up_proj = QLinearLayer(originalMLP.up_proj, args)
        self.act_fn = originalMLP.act_fn
        self.act_quant = Quantizer(args=args)

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        for name in ("gate_proj", "down_proj", "up_proj", "act_quant"):
            setattr(self, name, getattr(self, name).to(*args, **kwargs))
        return self

    def _act_is_silu(self):
        f = self.act_fn
        return isinstance(f, nn.SiLU) or type(f).__name__ in ("SiLUActivation", "SiLU") or f is nn.functional.silu

    FUSED_MIN_ROWS = 512          # the fused launch always runs the 256x256 geometry

    def _fused_gate_up(self):
        """The interleaved gate/up weight operand of gate_up_silu_quant_f6, cached per packed form of the two layers.  Built from the
        packed INT4 codes of both layers; a layer with keep_packed_with_f6 = False gets them re-packed for the build only and
        releases them again afterwards (the cache check itself does not need them)."""
        if self.gate_proj.bias is not None or self.up_proj.bias is not None:
            return None
        if self.gate_proj.packed_weight(need_codes=False) is None or self.up_proj.packed_weight(need_codes=False) is None:
            return None
        key = (self.gate_proj._packed_key, self.up_proj._packed_key)
        if not (self.gate_proj.keep_f6 and self.up_proj.keep_f6):    # 4-bit weights only (QLinearLayer.keep_f6): built per call, not kept
            self._fused = None
            return _ops.fuse_gate_up_weights(self.gate_proj.packed_weight(), self.up_proj.packed_weight())
        if getattr(self, "_fused", None) is None or self._fused[0] != key:
            pg, pu = self.gate_proj.packed_weight(), self.up_proj.packed_weight()
            self._fused = (key, _ops.fuse_gate_up_weights(pg, pu))
            self.gate_proj.release_codes()
            self.up_proj.release_codes()
        return self._fused[1]

    def _fused_is_bit_identical(self, rows):
        """The fused launch sums the K steps in order (256x256 geometry).  The stand-alone gate / up GEMMs do so too where
        atom_gemm_w4a4_f6_order(M, N_inter, K) == 1; for few-tile shapes they add two / four ordered K ranges instead, and the fp16
        gate / up values can differ in the last bit.  Fuse only where the results are bit-identical to the three launches, so that
        the model's output does not depend on which side of FUSED_MIN_ROWS a batch falls."""
        n, k = self.gate_proj.weight.shape
        return _ops.L.lib().atom_gemm_w4a4_f6_order(int(rows), int(n), int(k)) == 1

    @torch.no_grad()
    def forward(self, x):
        inter = self.gate_proj.weight.shape[0]
        hot = self.act_quant.hot_args(inter)
        codes = get_codes(x)
        if (hot is not None and codes is not None and codes.wide == "f6" and codes.rows >= self.FUSED_MIN_ROWS and x.is_cuda
                and self._act_is_silu() and codes.hidden == self.gate_proj.weight.shape[1] and inter % 128 == 0
                and self._fused_is_bit_identical(codes.rows)):
            fused = self._fused_gate_up()
            if fused is not None:
                # gate_proj, up_proj, act_fn(gate) * up and the quantiser in one launch: bit-identical to the path below
                o8, o6, s8, s4, xq = _ops.gate_up_silu_quant_f6(codes.o4, codes.o8, codes.s8, fused, quant_mode="sim",
                                                                clip=float(hot.a_clip_ratio), scale_layout=codes.layout,
                                                                return_dequant=True)
                act = attach_codes(xq.view(*x.shape[:-1], inter), ActCodes(o8, o6, s8, s4, codes.rows, inter, layout=codes.layout,
                                                                          wide="f6"))
                return self.down_proj(act)
        gate = self.gate_proj(x)
        up = self.up_proj(x)
        inter = gate.shape[-1]
        hot = self.act_quant.hot_args(inter)
        if hot is not None and gate.is_cuda and gate.dtype == torch.float16 and self._act_is_silu():
            shape = gate.shape
            g2, u2 = gate.reshape(-1, inter), up.reshape(-1, inter)
            wide = want_wide_codes(g2.shape[0])
            o8, o4, s8, s4, xq = _ops.activate_fp16_i4(g2, u2, quant_mode="sim", clip=float(hot.a_clip_ratio),
                                                       scale_layout="plain", return_dequant=True, wide_codes=wide)
            act = attach_codes(xq.view(shape), ActCodes(o8, o4, s8, s4, g2.shape[0], inter, wide=wide))
        else:
            act = self.act_quant(self.act_fn(gate) * up)
        return self.down_proj(act)
