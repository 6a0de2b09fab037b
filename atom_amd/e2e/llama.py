"""``LinearInt4`` / ``LlamaRMSNormInt4`` / ``LlamaMLP`` over atom_amd.ops -- the reference's real-kernel call graph
(e2e/punica-atom/punica/models/llama.py:35-87, 233-244) with the same parameter names, shapes and dtypes, so a state
dict written for the reference classes loads here.  Kernel-flavoured arithmetic and the replicated scale layout, i.e.
exactly what the reference's CUDA ops compute.  ``LlamaAttention`` / ``LlamaDecoderLayer`` (reference llama.py:90-292)
run the INT4 paged-KV ops of csrc/kv_i4.hip.
"""
from __future__ import annotations

import torch
from torch import nn

import dataclasses
import math

from .. import ops
from ..utils import BatchedKvCacheInt4, BatchLenInfo

GROUP = 128


@dataclasses.dataclass
class DecodeFusion:
    """Which launches of a decode step are fused -- a constructor argument of the modules below (``fusion=``); every module that gets
    none shares ``FUSION``, the process-wide default.  (Rounds 2-4 read these switches from environment variables at import.)
      decode     the projections that share an activation in ONE launch (q / k / v, gate / up), the residual add inside down_proj's
                 (False: one launch per projection, the reference's call order llama.py:259-292)
      kv_append  k / v quantised into the paged cache from the FP32 sums (False: the reference's op sequence _o4 GEMM -> append)
      kv_in_decode  ... by the attention launch itself (atom_batch_decode_append_i4, round 6; False: a launch of its own in front)
      merge_in_o_proj  one or two tokens whose KV range is split over several waves: the split merge runs in front of o_proj's quantiser
                 inside o_proj's launch (atom_gemm_w4a4_multi_merge_q, round 6) instead of the decode op's own merge launch:
                 bit-identical, a launch fewer.  Every one of o_proj's 256 workgroups re-reads the partial states of all heads (133 KB at 8
                 states per head: 2 k cycles of a CU's 64 B / clock), so the first form LOST (53.4 -> 57.2 us cold per layer with 16
                 states, 55.2 -> 56.6 with 8: profiles/r06/ab_merge_in_o_proj.txt); with the two-level merge in the attention launch
                 (4 states at context 1024: 66 KB), quantiser / streamer roles for this op and one (m, d) request per lane it is ahead --
                 46.3-46.8 -> 45.6-46.0 us at batch 1 (ab_merge_in_o_proj4.txt; two tokens of hidden <= 4096 run the decode-batch kernel and never
                 reach the op) -- and on by default.
      q_decode   one or two tokens: quantisers inside the GEMM that consumes them (atom_gemm_w4a4_multi_q)
      q_mask     ... which of the four (LlamaDecoderLayer._decode_fused_q): 1 input_layernorm -> q / k / v, 2 reorder -> o_proj,
                 4 add + post_attention_layernorm -> gate / up, 8 SiLU x up -> down_proj.  Default 15 since round 6: the quantiser runs once per
                 CU in front of the dot-product kernel (csrc/gemvq_w4a4.hip) and every one of the four pays -- a Llama-7B layer at batch 1,
                 cold: 60.8 us with the four quantisers as launches of their own, 5x.x us with them inside (profiles/r06/decode_layer_hot_cold.txt).
                 (Rounds 3-5 ran it in every 16-feature workgroup of the decode-batch kernel: only reorder -> o_proj paid, default 2.)
      q_mask2    the same choice for a step of TWO tokens: their projections with K <= 4096 run the decode-batch kernel (two tokens double
                 the dot-product kernel's arithmetic per weight chunk), where only reorder -> o_proj pays: default 2 (66-67 us per layer
                 cold; 15 measures 69-71)"""
    decode: bool = True
    kv_append: bool = True
    kv_in_decode: bool = True
    merge_in_o_proj: bool = True
    q_decode: bool = True
    q_mask: int = 15
    q_mask2: int = 2


FUSION = DecodeFusion()


class LinearInt4(nn.Module):
    """reference llama.py:35-68.  ``forward`` takes the 4-tuple an activation-quant op returns."""

    def __init__(self, in_features, out_features, out_dtype, bias=False, fusion: DecodeFusion = None):
        super().__init__()
        assert bias is False
        assert out_dtype in ("fp16", "int4")
        self.fusion = fusion if fusion is not None else FUSION
        self.in_features, self.out_features, self.out_dtype = in_features, out_features, out_dtype
        self.weight_int4 = nn.Parameter(torch.empty(out_features, (in_features - GROUP) // 2, dtype=torch.uint8),
                                        requires_grad=False)
        self.weight_int8 = nn.Parameter(torch.empty(out_features, GROUP, dtype=torch.int8), requires_grad=False)
        # over-allocated exactly like the reference (scale_size(out_features) columns); the GEMM reads it flat as [G][N]
        self.scale_int4 = nn.Parameter(torch.empty((in_features // GROUP - 1, ops.scale_size(out_features)),
                                                   dtype=torch.float16), requires_grad=False)
        self.scale_int8 = nn.Parameter(torch.empty(ops.scale_size(out_features), dtype=torch.float16),
                                       requires_grad=False)
        self.register_parameter("bias", None)

    @torch.no_grad()
    def load_fp16_weight(self, weight: torch.Tensor, w_clip: float = 0.85, channel_group: int = 2):
        """Fill the packed parameters from a (column-reordered) fp16 weight [out, in] -- the packer the reference lacks."""
        b4, b8, sb, sb8 = ops.quant_weight_w4(weight.contiguous(), w_clip, channel_group)
        g, n = sb.shape
        self.weight_int4.data = b4
        self.weight_int8.data = b8
        flat = torch.zeros(self.scale_int4.numel(), dtype=torch.float16, device=weight.device)
        flat[: g * n] = sb.reshape(-1)
        self.scale_int4.data = flat.view(self.scale_int4.shape)
        s8 = torch.zeros(self.scale_int8.numel(), dtype=torch.float16, device=weight.device)
        s8[:n] = sb8
        self.scale_int8.data = s8
        return self

    def packed(self):
        """(b4, b8, sb [G, N], sb8 [N]) views of the parameters (the scale parameters are over-allocated like the reference's)."""
        n, g = self.out_features, self.in_features // GROUP - 1
        return (self.weight_int4.data, self.weight_int8.data, self.scale_int4.data.reshape(-1)[: g * n].view(g, n),
                self.scale_int8.data.reshape(-1)[:n])

    def weight_f6s(self):
        """The weight in the F6 format with appended fp32 scales, made once per packed weight (prefill batches in the F6 format)."""
        key = (self.weight_int4.data_ptr(), self.weight_int4._version, self.scale_int4.data_ptr(), self.scale_int4._version)
        if getattr(self, "_f6s", None) is None or self._f6s[0] != key:
            b4, _, sb, _ = self.packed()
            self._f6s = (key, ops.repack_weight_f6(b4, sb.contiguous()))
        return self._f6s[1]

    def forward(self, input):
        outlier, norms, outlier_scales, norm_scales = input
        if norms.dim() == 2 and self.out_dtype == "fp16" and ops.gemm_recodes_cached(outlier.size(0), self.out_features, self.in_features):
            # a batch in the reference's packed format that the BF6 kernels serve faster once the weight's BF6 form exists (from 129 rows; from
            # 17 where the decode-batch kernel does not take the shape): re-code the activation here and use the layer's own cached BF6
            # weight (same kernel, same bits as the workspace route of the GEMM op, which would keep a second copy of the weight)
            norms = ops.repack_act_f6(norms.view(torch.uint8), norm_scales)
        if norms.dim() == 3:                                  # the F6 activation operand [G][rows_pad][104] (fp16 output only)
            assert self.out_dtype == "fp16"
            return ops.dense_layer_gemm_i4_fp16(norms, self.weight_f6s(), norm_scales, self.scale_int4, outlier, self.weight_int8,
                                                outlier_scales, self.scale_int8, a_wide="f6")
        f = {"int4": ops.dense_layer_gemm_i4_o4, "fp16": ops.dense_layer_gemm_i4_fp16}[self.out_dtype]
        return f(norms, self.weight_int4, norm_scales, self.scale_int4, outlier, self.weight_int8, outlier_scales,
                 self.scale_int8)

    def single(self):
        """this projection as an operand of dense_layer_gemm_i4_multi / _multi_q (one segment)"""
        key = ops.fused_key([self])
        if getattr(self, "_single", None) is None or self._single["key"] != key:
            b4, b8, sb, sb8 = self.packed()
            self._single = {"b4": b4.view(torch.uint8), "b8": b8, "sb": sb.contiguous(), "sb8": sb8.contiguous(), "n_seg": self.out_features,
                            "nseg": 1, "k": self.in_features, "key": key}
        return self._single

    def forward_add(self, input, residual):
        """residual + forward(input): decode batches run the add inside the projection's launch (same bits as the torch add)."""
        outlier, norms, outlier_scales, norm_scales = input
        rows = outlier.size(0)
        if (norms.dim() == 2 and self.out_dtype == "fp16" and self.fusion.decode and residual.is_contiguous()
                and ops.multi_gemm_fits(rows, self.out_features, 1, self.in_features)):
            return ops.dense_layer_gemm_i4_multi(norms, norm_scales, outlier, outlier_scales, self.single(),
                                                 add=residual.view(rows, self.out_features))[0].view(residual.shape)
        return residual + self.forward(input)

    def forward_f32(self, input):
        """FP32 sums (decode batches only): what the "int4" epilogue would quantise; see LlamaAttention.forward."""
        outlier, norms, outlier_scales, norm_scales = input
        return ops.dense_layer_gemm_i4_f32(norms, self.weight_int4, norm_scales, self.scale_int4, outlier, self.weight_int8,
                                           outlier_scales, self.scale_int8)


class LlamaRMSNormInt4(nn.Module):
    """reference llama.py:233-244."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=torch.float16), requires_grad=False)
        self.variance_epsilon = eps
        self.reorder_index = nn.Parameter(torch.randperm(hidden_size).to(torch.int16), requires_grad=False)

    def forward(self, hidden_states):
        return ops.rmsnorm_fp16_i4(hidden_states, self.weight, self.reorder_index, self.variance_epsilon)

    def forward_add(self, x, residual):
        """(x + residual, quantised RMSNorm of it) in one kernel -- the residual add of reference llama.py:268-275 fused."""
        out = ops.add_rmsnorm_fp16_i4(x, residual, self.weight, self.reorder_index, self.variance_epsilon)
        return out[0], out[1:]


class LlamaMLP(nn.Module):
    """reference llama.py:71-87: down( activate_fp16_i4( gate(x), up(x) ) )."""

    def __init__(self, config, fusion: DecodeFusion = None):
        super().__init__()
        self.config = config
        self.fusion = fusion if fusion is not None else FUSION
        self.hidden_size = config.hidden_size
        self.intermediate_size = config.intermediate_size
        self.gate_proj = LinearInt4(self.hidden_size, self.intermediate_size, out_dtype="fp16", bias=False, fusion=fusion)
        self.up_proj = LinearInt4(self.hidden_size, self.intermediate_size, out_dtype="fp16", bias=False, fusion=fusion)
        self.down_proj = LinearInt4(self.intermediate_size, self.hidden_size, out_dtype="fp16", bias=False, fusion=fusion)

    FUSED_MIN_ROWS = 512          # the fused launch always runs the 256x256 geometry

    def _fused_gate_up(self):
        key = tuple(t.data_ptr() for t in (self.gate_proj.weight_int4, self.up_proj.weight_int4)) + \
            tuple(t._version for t in (self.gate_proj.weight_int4, self.up_proj.weight_int4, self.gate_proj.scale_int4, self.up_proj.scale_int4))
        if getattr(self, "_fused", None) is None or self._fused[0] != key:
            self._fused = (key, ops.fuse_gate_up_weights(self.gate_proj.packed(), self.up_proj.packed()))
        return self._fused[1]

    def _decode_gate_up(self):
        """gate_proj + up_proj as one operand of dense_layer_gemm_i4_multi (decode batches)."""
        mods = [self.gate_proj, self.up_proj]
        if getattr(self, "_gu", None) is None or self._gu["key"] != ops.fused_key(mods):
            self._gu = ops.fuse_projection_weights(mods)
            self._fused = None                                   # (the prefill operand is keyed by the parameters' storage too)
        return self._gu

    def forward(self, x, residual=None):
        """``residual`` (decode layers pass it): returns residual + mlp(x), the add inside down_proj's launch for decode batches."""
        outlier, norms, outlier_scales, norm_scales = x
        rows = outlier.size(0)
        down = (lambda a: self.down_proj(a)) if residual is None else (lambda a: self.down_proj.forward_add(a, residual))
        if (rows >= self.FUSED_MIN_ROWS and self.intermediate_size % GROUP == 0
                and ops.L.lib().atom_gemm_w4a4_f6_order(rows, self.intermediate_size, self.hidden_size) == 1):
            # prefill batches: gate_proj + up_proj + activate_fp16_i4 in ONE launch, bit-identical to the three
            # launches below wherever those sum the K steps in order too (atom_gemm_w4a4_f6_order == 1: every shape that fills the
            # chip; few-tile shapes keep the three launches so that the output does not depend on the route);
            # the activation is re-coded to the F6 operand once (the two GEMMs would each do it in their workspace)
            a6 = norms if norms.dim() == 3 else ops.repack_act_f6(norms.view(torch.uint8), norm_scales)
            act = ops.gate_up_silu_quant_f6(a6, outlier, outlier_scales, self._fused_gate_up())
            return down(act)
        if norms.dim() == 2 and self.fusion.decode and ops.multi_gemm_fits(rows, self.intermediate_size, 2, self.hidden_size):
            # decode batches: gate_proj and up_proj in one launch (they read the same activation operand)
            gate, up = ops.dense_layer_gemm_i4_multi(norms, norm_scales, outlier, outlier_scales, self._decode_gate_up())
            return down(ops.activate_fp16_i4(gate, up))
        return down(ops.activate_fp16_i4(self.gate_proj(x), self.up_proj(x)))


def dequant_kv_u4(packed: torch.Tensor, param: torch.Tensor) -> torch.Tensor:
    """[T, heads, 64] u8 + [T, heads, 2] fp16 (scale, zero) -> [T, heads, 128] fp32: nibble * scale - zero, element 2j in
    the low nibble (quantization.cuh:59-84)."""
    u = torch.stack([packed & 0xF, packed >> 4], dim=-1).reshape(*packed.shape[:-1], packed.shape[-1] * 2).float()
    p = param.float()
    return u * p[..., 0:1] - p[..., 1:2]


def rope_llama(x: torch.Tensor, pos: torch.Tensor, theta: float = 1e4) -> torch.Tensor:
    """x [T, heads, 128] fp32, pos [T]: pairs (i, i + 64), freq_i = theta^(-2i/128) (decode.cuh:40-72)."""
    d = x.shape[-1]
    inv = theta ** (-torch.arange(0, d, 2, device=x.device, dtype=torch.float32) / d)
    ang = pos.to(torch.float32)[:, None] * inv[None, :]
    cos = torch.cat([ang.cos(), ang.cos()], -1)[:, None, :]
    sin = torch.cat([ang.sin(), ang.sin()], -1)[:, None, :]
    rot = torch.cat([-x[..., d // 2:], x[..., :d // 2]], -1)
    return x * cos + rot * sin


def _append_and_decode(fusion, q, k32, v32, decode_kv, layer_idx, rope_theta, merge=True):
    """reference llama.py:168-196 for a pure decode step: this token's k / v into the INT4 paged cache, then attention over it -- one
    launch (DecodeFusion.kv_in_decode) or two; the same cache bytes and the same output either way.  ``merge=False``: the KV-split
    partial states instead of the output (ops.batch_decode_i4)."""
    if fusion.kv_in_decode:
        return ops.batch_decode_i4(q, decode_kv, layer_idx, rope_theta=rope_theta, append_kv=(k32, v32), merge=merge)
    ops.quant_append_kv_i4(decode_kv, k32, v32, layer_idx)
    return ops.batch_decode_i4(q, decode_kv, layer_idx, rope_theta=rope_theta, merge=merge)


class LlamaAttention(nn.Module):
    """reference llama.py:90-230: q/k/v projections (k, v with the u4 epilogue), KV written to the INT4 paged cache,
    decode requests through the RoPE-fused batch-decode kernel, output reordered + quantised + projected.
    Prefill requests: the reference attends to RANDOM keys and values there (llama.py:164-167, "HACK": a latency
    harness); this class attends to the de-quantised projections it has just written to the cache (causal, RoPE at
    positions 0..len-1) -- the values a later decode step reads back."""

    def __init__(self, config, layer_idx: int, fusion: DecodeFusion = None):
        super().__init__()
        self.config = config
        self.fusion = fusion if fusion is not None else FUSION
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self._scale = 1 / math.sqrt(self.head_dim)
        self.layer_idx = layer_idx
        if self.head_dim * self.num_heads != self.hidden_size:
            raise ValueError(f"hidden_size {self.hidden_size} is not divisible by num_heads {self.num_heads}")
        if self.head_dim != 128:
            raise ValueError("the INT4 KV kernels are built for head_dim 128 (as the reference's, punica_ops.cc:112)")
        self.q_proj = LinearInt4(self.hidden_size, self.hidden_size, out_dtype="fp16", bias=False, fusion=fusion)
        self.k_proj = LinearInt4(self.hidden_size, self.hidden_size, out_dtype="int4", bias=False, fusion=fusion)
        self.v_proj = LinearInt4(self.hidden_size, self.hidden_size, out_dtype="int4", bias=False, fusion=fusion)
        self.o_proj = LinearInt4(self.hidden_size, self.hidden_size, out_dtype="fp16", bias=False, fusion=fusion)
        self.reorder_index = nn.Parameter(torch.randperm(self.hidden_size).to(torch.int16), requires_grad=False)
        self.rope_theta = float(getattr(config, "rope_theta", 1e4))

    def _decode_qkv(self):
        """q_proj + k_proj + v_proj as one operand of dense_layer_gemm_i4_multi (decode steps)."""
        mods = [self.q_proj, self.k_proj, self.v_proj]
        if getattr(self, "_qkv", None) is None or self._qkv["key"] != ops.fused_key(mods):
            self._qkv = ops.fuse_projection_weights(mods)
        return self._qkv

    def forward(self, hidden_states, blen: BatchLenInfo, prefill_kv: BatchedKvCacheInt4 | None,
                decode_kv: BatchedKvCacheInt4 | None) -> torch.Tensor:
        nh, hd = self.num_heads, self.head_dim
        rows = hidden_states[0].size(0)
        pure_decode = (len(blen.prefills) == 0 and blen.decode == rows and self.fusion.kv_append
                       and ops.decode_gemm_fits(rows, self.hidden_size, self.hidden_size))
        fuse_qkv = (pure_decode and self.fusion.decode and hidden_states[1].dim() == 2
                    and ops.multi_gemm_fits(rows, self.hidden_size, 3, self.hidden_size))
        q_proj = None if fuse_qkv else self.q_proj(hidden_states)
        if pure_decode:
            # pure decode step: k / v sums in FP32, then ONE launch quantises both per head and writes the cache slots
            # (same cache contents as the u4-epilogue GEMMs + append_kv_i4 below; three launches fewer)
            assert decode_kv is not None
            if q_proj is None:                                   # q / k / v from one launch: q fp16, k and v as FP32 sums
                outlier, norms, outlier_scales, norm_scales = hidden_states
                q_proj, k32, v32 = ops.dense_layer_gemm_i4_multi(norms, norm_scales, outlier, outlier_scales, self._decode_qkv(),
                                                                 f32_mask=0b110)
            else:
                k32, v32 = self.k_proj.forward_f32(hidden_states), self.v_proj.forward_f32(hidden_states)
            o = _append_and_decode(self.fusion, q_proj.view(rows, nh, hd), k32, v32, decode_kv, self.layer_idx, self.rope_theta)
            return self.o_proj(ops.reorder_fp16_i4(o.view(rows, self.hidden_size), self.reorder_index))
        k_u4, k_sz = self.k_proj(hidden_states)
        v_u4, v_sz = self.v_proj(hidden_states)
        outs = []
        if len(blen.prefills) > 0:
            assert prefill_kv is not None
            k = k_u4[:blen.doff].view(-1, nh, hd // 2)
            v = v_u4[:blen.doff].view(-1, nh, hd // 2)
            ks = k_sz[:blen.doff].view(-1, nh, 2)
            vs = v_sz[:blen.doff].view(-1, nh, 2)
            ops.init_kv_i4(prefill_kv, k, v, ks, vs, blen.indptr, self.layer_idx)
            kf, vf = dequant_kv_u4(k, ks), dequant_kv_u4(v, vs)
            beg = 0
            for q_len in blen.prefills:
                sl = slice(beg, beg + q_len)
                pos = torch.arange(q_len, device=q_proj.device)
                q = rope_llama(q_proj[sl].view(q_len, nh, hd).float(), pos, self.rope_theta).transpose(0, 1)
                kk = rope_llama(kf[sl], pos, self.rope_theta).transpose(0, 1)
                o = torch.nn.functional.scaled_dot_product_attention(q, kk, vf[sl].transpose(0, 1), is_causal=True)
                outs.append(o.transpose(0, 1).reshape(q_len, self.hidden_size).to(q_proj.dtype))
                beg += q_len
        if blen.decode > 0:
            assert decode_kv is not None
            q = q_proj[blen.doff:].view(blen.decode, nh, hd)
            ops.append_kv_i4(decode_kv, k_u4[blen.doff:].view(blen.decode, nh, hd // 2),
                             v_u4[blen.doff:].view(blen.decode, nh, hd // 2), k_sz[blen.doff:].view(blen.decode, nh, 2),
                             v_sz[blen.doff:].view(blen.decode, nh, 2), self.layer_idx)
            o = ops.batch_decode_i4(q.contiguous(), decode_kv, self.layer_idx, rope_theta=self.rope_theta)
            outs.append(o.view(blen.decode, self.hidden_size))
        attn = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        return self.o_proj(ops.reorder_fp16_i4(attn.contiguous(), self.reorder_index))


class LlamaDecoderLayer(nn.Module):
    """reference llama.py:247-292."""

    def __init__(self, config, layer_idx: int, fusion: DecodeFusion = None):
        super().__init__()
        self.fusion = fusion if fusion is not None else FUSION
        self.hidden_size = config.hidden_size
        self.self_attn = LlamaAttention(config=config, layer_idx=layer_idx, fusion=fusion)
        self.mlp = LlamaMLP(config, fusion=fusion)
        self.input_layernorm = LlamaRMSNormInt4(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = LlamaRMSNormInt4(config.hidden_size, eps=config.rms_norm_eps)

    def _decode_fused_q(self, hidden_states, decode_kv, mask):
        """One or two tokens, pure decode: quantisers inside the GEMM that consumes their output (atom_gemm_w4a4_multi_q), per bit of
        ``mask`` -- 1 input_layernorm -> q / k / v, 2 reorder -> o_proj, 4 residual add + post_attention_layernorm -> gate / up,
        8 SiLU x up -> down_proj; the same bits as separate launches either way (reference call order llama.py:259-292)."""
        at, mlp = self.self_attn, self.mlp
        rows, hs = hidden_states.shape
        il, pl = self.input_layernorm, self.post_attention_layernorm
        if mask & 1:
            (q, k32, v32), _ = ops.dense_layer_gemm_i4_multi_q("rmsnorm", hidden_states, at._decode_qkv(), x2=il.weight, reorder_index=il.reorder_index,
                                                               eps=il.variance_epsilon, f32_mask=0b110)
        else:
            outlier, norms, outlier_scales, norm_scales = il(hidden_states)
            q, k32, v32 = ops.dense_layer_gemm_i4_multi(norms, norm_scales, outlier, outlier_scales, at._decode_qkv(), f32_mask=0b110)
        splits = ops.decode_splits(rows, decode_kv) if (mask & 2) and self.fusion.merge_in_o_proj else 1
        if splits >= 2 and ops.merge_q_gemm_fits(rows, hs, 1, hs, splits):
            # the KV-split merge in front of o_proj's quantiser, inside o_proj's launch (round 6): same bits, a launch fewer
            part = _append_and_decode(self.fusion, q.view(rows, at.num_heads, at.head_dim), k32, v32, decode_kv, at.layer_idx, at.rope_theta, merge=False)
            (attn,) = ops.dense_layer_gemm_i4_merge_q(part, splits, at.o_proj.single(), reorder_index=at.reorder_index)
        else:
            o = _append_and_decode(self.fusion, q.view(rows, at.num_heads, at.head_dim), k32, v32, decode_kv, at.layer_idx, at.rope_theta).view(rows, hs)
            if mask & 2:
                (attn,), _ = ops.dense_layer_gemm_i4_multi_q("reorder", o, at.o_proj.single(), reorder_index=at.reorder_index)
            else:
                attn = at.o_proj(ops.reorder_fp16_i4(o, at.reorder_index))
        if mask & 4:
            (gate, up), residual = ops.dense_layer_gemm_i4_multi_q("add_rmsnorm", attn, mlp._decode_gate_up(), residual=hidden_states, x2=pl.weight,
                                                                   reorder_index=pl.reorder_index, eps=pl.variance_epsilon)
        else:
            residual, (outlier, norms, outlier_scales, norm_scales) = pl.forward_add(attn, hidden_states)
            gate, up = ops.dense_layer_gemm_i4_multi(norms, norm_scales, outlier, outlier_scales, mlp._decode_gate_up())
        if mask & 8:
            (out,), _ = ops.dense_layer_gemm_i4_multi_q("silu_mul", gate, mlp.down_proj.single(), x2=up, add=residual)
            return out
        return mlp.down_proj.forward_add(ops.activate_fp16_i4(gate, up), residual)

    def _q_mask_for(self, rows):
        return self.fusion.q_mask if rows <= 1 else self.fusion.q_mask2

    def _fused_q_fits(self, rows):
        """every projection of the layer is a shape atom_gemm_w4a4_multi_q takes at this batch size (asked once per batch size)"""
        ok = getattr(self, "_fq_ok", None)
        if ok is None:
            ok = self._fq_ok = {}
        if rows not in ok:
            hs, inter = self.hidden_size, self.mlp.intermediate_size
            # only the ops the mask selects have to fit (each with ITS quantiser's bounds); the others run as separate launches
            need = (("rmsnorm", hs, 3, hs), ("reorder", hs, 1, hs), ("add_rmsnorm", inter, 2, hs), ("silu_mul", hs, 1, inter))
            ok[rows] = all(ops.multi_q_gemm_fits(q, rows, n, nseg, k) for bit, (q, n, nseg, k) in enumerate(need) if (self._q_mask_for(rows) >> bit) & 1)
            # the launches that take the un-fused operands in _decode_fused_q
            ok[rows] = ok[rows] and ops.multi_gemm_fits(rows, hs, 3, hs) and ops.multi_gemm_fits(rows, inter, 2, hs) and ops.multi_gemm_fits(rows, hs, 1, inter)
        return ok[rows]

    def forward(self, hidden_states, blen: BatchLenInfo, prefill_kv, decode_kv) -> torch.Tensor:
        rows = hidden_states.size(0) if torch.is_tensor(hidden_states) else 0
        fu = self.fusion
        if (fu.q_decode and self._q_mask_for(rows) and fu.decode and fu.kv_append and 0 < rows <= 2 and hidden_states.dim() == 2 and hidden_states.is_contiguous()
                and len(blen.prefills) == 0 and blen.decode == rows and decode_kv is not None and self._fused_q_fits(rows)):
            return self._decode_fused_q(hidden_states, decode_kv, self._q_mask_for(rows))
        attn = self.self_attn(self.input_layernorm(hidden_states), blen, prefill_kv, decode_kv)
        residual, normed = self.post_attention_layernorm.forward_add(attn, hidden_states)   # fused residual add
        return self.mlp(normed, residual=residual)                                           # ... and the second one (decode: in down_proj's launch)


class LlamaRMSNorm(nn.Module):
    """The final, un-quantised norm (reference llama.py:316 takes transformers' LlamaRMSNorm: fp32 statistics, weight applied after
    the cast back)."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=torch.float16), requires_grad=False)
        self.variance_epsilon = eps

    def forward(self, x):
        h = x.float()
        h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + self.variance_epsilon)
        return self.weight * h.to(x.dtype)


class LlamaModel(nn.Module):
    """reference llama.py:306-340: embedding -> decoder layers -> final norm over a flat [tokens, hidden] batch (prefill requests
    first, then one row per decode request).  Plain ``nn.Module`` (the reference derives from transformers' PreTrainedModel only for
    ``from_pretrained``; here ``load_state_dict(model.export.load_packed(path))`` fills it) and every layer gets its own
    ``layer_idx`` (the reference passes 0 to all of them, "Hack for memory", :313-314)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, getattr(config, "pad_token_id", None), dtype=torch.float16)
        self.layers = nn.ModuleList([LlamaDecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    @torch.no_grad()
    def forward(self, input_ids, blen: BatchLenInfo, prefill_kv, decode_kv) -> torch.Tensor:
        hidden_states = self.embed_tokens(input_ids)
        for layer in self.layers:
            hidden_states = layer(hidden_states, blen, prefill_kv, decode_kv)
        return self.norm(hidden_states)


class LlamaForCausalLM(nn.Module):
    """reference llama.py:343-364: returns (logits, hidden_states)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = LlamaModel(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False, dtype=torch.float16)

    @torch.no_grad()
    def forward(self, input_ids, blen: BatchLenInfo, prefill_kv, decode_kv):
        hidden_states = self.model(input_ids, blen, prefill_kv, decode_kv)
        return self.lm_head(hidden_states), hidden_states
