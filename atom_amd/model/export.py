"""On-disk form of a quantised model: the packed INT4/INT8 operands of every ``QLinearLayer`` plus the
reorder indices / norm weights, under the parameter names and shapes of the reference's real-kernel classes
(``LinearInt4.weight_int4 / weight_int8 / scale_int4 / scale_int8``, ``LlamaRMSNormInt4.weight / reorder_index``,
``LlamaAttention.reorder_index``; e2e/punica-atom/punica/models/llama.py:35-58,105-108,235-240), so the file loads into
``atom_amd.e2e`` modules -- or the reference's own -- with ``load_state_dict``.  The reference has no such bridge: its
simulated-quant tree (``model/``) and its kernels (``e2e/``) never meet.

    sd = packed_state_dict(model)            # after quantize_model_llama / quantize_model_gptq_llama
    save_packed(model, "llama-7b-w4a4.safetensors");  e2e_model.load_state_dict(load_packed(path), strict=False)
"""
from __future__ import annotations

import torch

from .. import ops as _ops
from .qLinearLayer import QLinearLayer


def _linear_entries(layer: QLinearLayer):
    packed = layer.packed_weight()
    if packed is None:
        raise ValueError("layer has no packed INT4 form (not quantised in the W4A4-g128 configuration, or not on a GPU)")
    b4, b8, sb, sb8 = packed
    g, n = sb.shape
    ld = _ops.scale_size(n)                   # the reference over-allocates both scale tensors (llama.py:49-55)
    s4 = torch.zeros(g * ld, dtype=torch.float16, device=sb.device)
    s4[: g * n] = sb.reshape(-1)              # ... and its kernel reads them flat as [G][N] (Dense_layer_gemm_i4_o16.cuh:497)
    s8 = torch.zeros(ld, dtype=torch.float16, device=sb.device)
    s8[:n] = sb8
    return {"weight_int4": b4, "weight_int8": b8, "scale_int4": s4.view(g, ld), "scale_int8": s8}


@torch.no_grad()
def packed_state_dict(module: torch.nn.Module, prefix: str = "") -> dict:
    out = {}
    for name, mod in module.named_modules():
        full = f"{prefix}{name}"
        dot = f"{full}." if full else ""
        if type(mod) is QLinearLayer:
            if not mod.enable_quant:
                out[f"{dot}weight"] = mod.weight
                continue
            for k, v in _linear_entries(mod).items():
                out[f"{dot}{k}"] = v
            if mod.bias is not None:
                out[f"{dot}bias"] = mod.bias
        idx = getattr(mod, "reorder_index", None)
        if torch.is_tensor(idx):
            out[f"{dot}reorder_index"] = idx.to(torch.int16)
        norm = getattr(mod, "originalNorm", None)
        if norm is not None and hasattr(norm, "weight"):
            out[f"{dot}weight"] = norm.weight.detach().to(torch.float16)
    # what the quantisation flow leaves untouched (embedding, final norm, lm_head of a whole model): as they are.  Everything that lives
    # under a QLinearLayer / a wrapped norm is already there in its packed form; rotary tables are recomputed by the kernels.
    covered = tuple(f"{prefix}{name}." for name, mod in module.named_modules()
                    if type(mod) is QLinearLayer or hasattr(mod, "originalNorm") or "rotary_emb" in name.split("."))
    for k, v in module.state_dict().items():
        full = f"{prefix}{k}"
        if full not in out and not full.startswith(covered) and torch.is_tensor(v) and not k.endswith("reorder_index"):
            out[full] = v
    return {k: v.detach().contiguous().cpu() for k, v in out.items()}


def save_packed(module: torch.nn.Module, path: str, prefix: str = "") -> None:
    from safetensors.torch import save_file
    save_file(packed_state_dict(module, prefix), path, metadata={"format": "atom-w4a4-g128-keeper128", "version": "1"})


def load_packed(path: str, device: str = "cpu") -> dict:
    from safetensors.torch import load_file
    return load_file(path, device=device)
