"""HIP-path mirror of the reference's ``model/modelutils_llama.py`` (same function names, positional arguments and effects on
``model``), so that the reference's driver (``main.py:199-270``) runs on the MI355X path when ``atom_amd/dropin`` is first on
``sys.path``:

    reorder_model_llama(model, device, args, reorder_index)      reference :15-75
    add_act_quant_wrapper_llama(model, device, args, scales)     reference :77-124
    quantize_model_llama(model, device, args)                    reference :126-153
    quantize_model_gptq_llama(model, device, args, dataloader)   reference :155-266

What is different underneath.  The reference bounces every layer GPU -> CPU after touching it (24 GB cards); an MI355X holds a
whole Llama-70B in fp16 twice over, so the layers STAY on ``device`` by default (``offload=False``) and the packed INT4 / F6
operands that ``QLinearLayer.quant()`` produced there stay valid -- ``offload=True`` restores the reference's behaviour (the packed
form is then re-keyed when the layer comes back, ``QLinearLayer.to``).  The three passes are table-driven instead of spelled out
per projection.  The GPTQ solver itself (reference gptq.py:63-339) is not part of the W4A4 hot path and is not rebuilt here: the
calibration pass takes the reference's ``gptq`` module when it is importable, or any object with the same three-call protocol
(``solver(layer, n_out=, keeper_precision=)`` -> ``.add_batch(inp, out)``, ``.fasterquant(percdamp=, groupsize=)``, ``.free()``);
what the HIP path owns is everything around it: the hooks, ``layer.weight.data = Q`` -> ``atom_pack_weight_w4`` on the next
forward, and the W4A4 layer outputs that feed the next layer's Hessians.
"""
from __future__ import annotations

import gc
from functools import partial

import torch
from torch import nn

from .qLinearLayer import find_qlinear_layers
from .qLlamaLayer import QLlamaDecoderLayer
from .quant import quantize_activation_wrapper, quantize_attn_k_wrapper, quantize_attn_v_wrapper

# (module, projection, whose INPUT order permutes this projection's output rows) -- reference :29-61
_PROJECTIONS = (("mlp", "gate_proj", "down_proj"), ("mlp", "up_proj", "down_proj"), ("mlp", "down_proj", None),
                ("self_attn", "q_proj", None), ("self_attn", "k_proj", None), ("self_attn", "v_proj", None),
                ("self_attn", "o_proj", None))
# where the runtime gathers live: (owner of the buffer, the projection whose input order it is) -- reference :62-68
_GATHERS = (("input_layernorm", "self_attn", "k_proj"), ("post_attention_layernorm", "mlp", "gate_proj"), ("self_attn", "self_attn", "o_proj"))
# activation quantisers and the projection they feed (the key of `scales`) -- reference :96-120
_ACT_QUANT = (("self_attn", "self_attn", "o_proj"), ("mlp", "mlp", "down_proj"), ("input_layernorm", "self_attn", "k_proj"),
              ("post_attention_layernorm", "mlp", "gate_proj"))


def _is_plain_llama_layer(layer) -> bool:
    try:
        from transformers.models.llama.modeling_llama import LlamaDecoderLayer
    except Exception:                                     # pragma: no cover - transformers is a dependency of the reference flow
        return False
    return isinstance(layer, LlamaDecoderLayer)


def _wrapped(layer, args):
    """QLlamaDecoderLayer for a layer of the model: wrap a transformers layer, pass a wrapped one through, skip anything else."""
    if isinstance(layer, QLlamaDecoderLayer):
        return layer
    if _is_plain_llama_layer(layer):
        return QLlamaDecoderLayer(originalLayer=layer, args=args)
    return None


def _settle(layers, i, m, offload):
    layers[i] = m.cpu() if offload else m
    if offload and torch.cuda.is_available():
        torch.cuda.empty_cache()


def reorder_model_llama(model, device, args, reorder_index, offload: bool = False):
    """Permute every projection's input columns (and gate / up output rows) by the calibrated channel order and register the three
    runtime gather indices.  q / k / v take k_proj's order, as the reference does (:63-65)."""
    model.config.use_cache = False
    assert reorder_index is not None, "Reorder index is None"
    layers = model.model.layers
    for i in range(len(layers)):
        m = _wrapped(layers[i].to(device), args)
        if m is None:
            continue
        key = lambda mod, proj: reorder_index[f"layers.{i}.{mod}.{proj}.input"]
        for mod, proj, rows_from in _PROJECTIONS:
            getattr(getattr(m, mod), proj).reorder(in_reorder_index=key(mod, proj),
                                                   out_reorder_index=None if rows_from is None else key(mod, rows_from))
        for owner, mod, proj in _GATHERS:
            getattr(m, owner).register_buffer("reorder_index", key(mod, proj).to(device))
        _settle(layers, i, m, offload)
    return model


def add_act_quant_wrapper_llama(model, device, args, scales, offload: bool = False):
    """Configure the four activation quantisers and the K / V cache quantisers of every layer."""
    model.config.use_cache = False
    layers = model.model.layers
    act = partial(quantize_activation_wrapper, args=args)
    for i in range(len(layers)):
        m = _wrapped(layers[i], args)
        if m is None:
            continue
        m = m.to(device)
        for owner, mod, proj in _ACT_QUANT:
            getattr(m, owner).act_quant.configure(act, scales[f"layers.{i}.{mod}.{proj}"])
        m.self_attn.v_quant.configure(partial(quantize_attn_v_wrapper, args=args), None)
        m.self_attn.k_quant.configure(partial(quantize_attn_k_wrapper, args=args), None)
        _settle(layers, i, m, offload)
    return model


def quantize_model_llama(model, device, args, offload: bool = False):
    """Round-to-nearest weight quantisation: QLinearLayer.quant() of all seven projections (one HIP launch each, which also leaves the
    packed INT4 / INT8 operands of the W4A4 GEMM on ``device``)."""
    model.config.use_cache = False
    layers = model.model.layers
    for i in range(len(layers)):
        m = _wrapped(layers[i], args)
        if m is None:
            continue
        m = m.to(device)
        for mod, proj, _ in _PROJECTIONS:
            getattr(getattr(m, mod), proj).quant()
        _settle(layers, i, m, offload)
    return model


class _StopForward(Exception):
    pass


class _FirstLayerInputs(nn.Module):
    """Stands in for layer 0 while the calibration / evaluation batches go through the embedding: records what layer 0 would have
    received (the hidden states and the two keyword arguments the layers are later called with) and stops the forward
    (reference :172-184, eval.py:26-36 do this with a `Catcher` that raises ValueError)."""

    def __init__(self, module, store):
        super().__init__()
        self.module = module
        self.self_attn = getattr(module, "self_attn", None)
        self.store = store

    def forward(self, inp, **kwargs):
        s = self.store
        s["inps"][s["i"]] = inp
        s["i"] += 1
        s["attention_mask"] = kwargs.get("attention_mask")
        s["position_ids"] = kwargs.get("position_ids")
        s["position_embeddings"] = kwargs.get("position_embeddings")
        raise _StopForward


def capture_first_layer_inputs(model, batches, nsamples, device, offload: bool = False):
    """Run ``batches`` (an iterable of [1, seqlen] token tensors) up to layer 0 and return (inps [nsamples, seqlen, hidden],
    layer keyword arguments)."""
    layers = model.model.layers
    model.model.embed_tokens = model.model.embed_tokens.to(device)
    if getattr(model.model, "rotary_emb", None) is not None:
        model.model.rotary_emb = model.model.rotary_emb.to(device)
    dtype = next(iter(model.parameters())).dtype
    store = {"i": 0, "inps": torch.zeros((nsamples, model.seqlen, model.config.hidden_size), dtype=dtype, device=device)}
    layers[0] = _FirstLayerInputs(layers[0].to(device), store)
    try:
        for batch in batches:
            try:
                model(batch.to(device))
            except _StopForward:
                pass
    finally:
        layers[0] = layers[0].module
    if offload:
        layers[0] = layers[0].cpu()
        model.model.embed_tokens = model.model.embed_tokens.cpu()
    kw = {"attention_mask": store.get("attention_mask"), "position_ids": store.get("position_ids")}
    if store.get("position_embeddings") is not None:
        kw["position_embeddings"] = store["position_embeddings"]
    return store["inps"], kw


def _reference_solver():
    try:
        import gptq as _g                                   # the reference's model/gptq.py, when its directory is on sys.path
    except Exception as e:
        raise ImportError("quantize_model_gptq_llama needs a GPTQ solver: put the reference's model/ directory on sys.path (after "
                          "atom_amd/dropin) or pass solver=...") from e

    def make(layer, args):
        s = _g.GPTQ(layer, n_out=args.keeper, keeper_precision=args.keeper_precision)
        s.quantizer = _g.Quantizer_GPTQ()
        s.quantizer.configure(args.wbits, perchannel=True, sym=args.w_sym, mse=False, channel_group=args.weight_channel_group,
                              clip_ratio=args.w_clip_ratio, quant_type=args.quant_type)
        return s
    return make


@torch.no_grad()
def quantize_model_gptq_llama(model, device, args, dataloader, solver=None, offload: bool = False):
    """GPTQ calibration over the HIP layers (reference :155-266): per layer, Hessians from forward hooks on its QLinearLayers
    (``hook(_, inp, out)`` sees the fake-quantised fp16 activation ``inp[0]``, exactly what the reference's hook sees), the solver
    writes ``layer.weight.data = Q`` (gptq.py:331), and the layer is run once more -- now through ``atom_pack_weight_w4`` + the W4A4
    GEMM -- to produce the next layer's inputs.  ``solver(layer, args)`` returns the per-projection solver object; default = the
    reference's."""
    make = solver if solver is not None else _reference_solver()
    use_cache = model.config.use_cache
    model.config.use_cache = False
    layers = model.model.layers
    model.model.norm = model.model.norm.to(device)
    inps, kw = capture_first_layer_inputs(model, (b[0] for b in dataloader), args.nsamples, device, offload)
    if offload:
        model.model.norm = model.model.norm.cpu()
    outs = torch.zeros_like(inps)
    quantizers = {}
    for i in range(len(layers)):
        m = _wrapped(layers[i], args)
        if m is None:
            continue
        layer = m.to(device)
        subset = find_qlinear_layers(layer)
        solvers = {name: make(lin, args) for name, lin in subset.items()}
        handles = [lin.register_forward_hook(lambda _, inp, out, s=solvers[name]: s.add_batch(inp[0].data, out.data))
                   for name, lin in subset.items()]
        for j in range(args.nsamples):
            layer(inps[j].unsqueeze(0), **kw)
        for h in handles:
            h.remove()
        for name, s in solvers.items():
            s.fasterquant(percdamp=args.percdamp, groupsize=args.weight_group_size)
            q = getattr(s, "quantizer", None)
            quantizers[f"model.layers.{i}.{name}"] = q.cpu() if hasattr(q, "cpu") else q
            s.free()
        del solvers
        for j in range(args.nsamples):
            outs[j] = layer(inps[j].unsqueeze(0), **kw)[0]
        _settle(layers, i, layer, offload)
        del layer, m
        gc.collect()
        inps, outs = outs, inps
    model.config.use_cache = use_cache
    return model
