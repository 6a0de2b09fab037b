"""``LinearInt4`` / ``LlamaRMSNormInt4`` / ``LlamaMLP`` over atom_amd.ops -- the reference's real-kernel call graph
(e2e/punica-atom/punica/models/llama.py:35-87, 233-244) with the same parameter names, shapes and dtypes, so a state
dict written for the reference classes loads here.  Kernel-flavoured arithmetic and the replicated scale layout, i.e.
exactly what the reference's CUDA ops compute.  (Attention / paged INT4 KV: out of the hot-path scope, SURVEY 2a #12.)
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops

GROUP = 128


class LinearInt4(nn.Module):
    """reference llama.py:35-68.  ``forward`` takes the 4-tuple an activation-quant op returns."""

    def __init__(self, in_features, out_features, out_dtype, bias=False):
        super().__init__()
        assert bias is False
        assert out_dtype in ("fp16", "int4")
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

    def forward(self, input):
        outlier, norms, outlier_scales, norm_scales = input
        f = {"int4": ops.dense_layer_gemm_i4_o4, "fp16": ops.dense_layer_gemm_i4_fp16}[self.out_dtype]
        return f(norms, self.weight_int4, norm_scales, self.scale_int4, outlier, self.weight_int8, outlier_scales,
                 self.scale_int8)


class LlamaRMSNormInt4(nn.Module):
    """reference llama.py:233-244."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=torch.float16), requires_grad=False)
        self.variance_epsilon = eps
        self.reorder_index = nn.Parameter(torch.randperm(hidden_size).to(torch.int16), requires_grad=False)

    def forward(self, hidden_states):
        return ops.rmsnorm_fp16_i4(hidden_states, self.weight, self.reorder_index, self.variance_epsilon)


class LlamaMLP(nn.Module):
    """reference llama.py:71-87: down( activate_fp16_i4( gate(x), up(x) ) )."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.hidden_size = config.hidden_size
        self.intermediate_size = config.intermediate_size
        self.gate_proj = LinearInt4(self.hidden_size, self.intermediate_size, out_dtype="fp16", bias=False)
        self.up_proj = LinearInt4(self.hidden_size, self.intermediate_size, out_dtype="fp16", bias=False)
        self.down_proj = LinearInt4(self.intermediate_size, self.hidden_size, out_dtype="fp16", bias=False)

    def forward(self, x):
        return self.down_proj(ops.activate_fp16_i4(self.gate_proj(x), self.up_proj(x)))
