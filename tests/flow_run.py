"""Run the quantisation + evaluation FLOW of the reference's driver (main.py:224-270) on the tiny model of tests/flow_model.py and
record what the flow tests compare.  One file, three users:

  * tests/golden/gen_golden_flow.py  -- ``impl="reference"``: the UNMODIFIED /root/reference/model modules (modelutils_llama, eval, gptq,
    quant, qLinearLayer, qLlamaLayer), CPU, build container only -> tests/golden/flow_*.npz
  * tests/test_flow_reference_cpu.py -- ``impl="ref_flow_on_dropin"``: the UNMODIFIED reference modelutils_llama / eval / gptq driving
    OUR classes (atom_amd/dropin first on sys.path), CPU, configurations that need no kernel (16-bit activations)
  * tests/test_gpu_flow.py           -- ``impl="atom"``: atom_amd.model.modelutils_llama / eval on cuda:0 (nothing of /root/reference);
    ``impl="atom_cpu"`` (tests/test_flow_reference_cpu.py): the same mirrors on CPU for the 16-bit-activation configurations

Run as a script it executes one (impl, config) in a fresh interpreter (the reference's modules are top-level names -- ``quant``,
``qLinearLayer`` ... -- so two implementations cannot share a process) and writes an .npz.
"""
from __future__ import annotations

import argparse
import importlib
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/model"
ROW_STEP = 8          # every 8th token row of sample 0 is recorded

CONFIGS = {
    # name: (model kwargs, args overrides, seqlens to evaluate, gptq)
    "rtn_w4a4": (dict(hidden=512, heads=4, inter=1408, layers=2, vocab=1000), dict(), (96, 320), False),
    "gptq_w4a4": (dict(hidden=256, heads=2, inter=384, layers=2, vocab=1000), dict(), (96,), True),
    # grouped-query attention (4 query heads on 2 K/V heads: k_proj / v_proj are 512 -> 256, repeat_kv before the V quantiser)
    "rtn_w4a4_gqa": (dict(hidden=512, heads=4, kv_heads=2, inter=1408, layers=2, vocab=1000), dict(), (96,), False),
    "rtn_w4a16_gqa": (dict(hidden=512, heads=4, kv_heads=2, inter=1408, layers=2, vocab=1000), dict(abits=16), (96,), False),
    "rtn_w4a16": (dict(hidden=512, heads=4, inter=1408, layers=2, vocab=1000), dict(abits=16), (96,), False),
    "gptq_w4a16": (dict(hidden=256, heads=2, inter=384, layers=2, vocab=1000), dict(abits=16), (96,), True),
}


def _stub_dir():
    """Modules that are NOT ours and fail to import for unrelated reasons: bitsandbytes (not installed; quant.py:4 / gptq.py:15 import
    two names used only for --quant_type fp) and the reference's qMixtralLayer (needs transformers 4.39's Mixtral classes)."""
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "bitsandbytes"))
    open(os.path.join(d, "bitsandbytes", "__init__.py"), "w").close()
    with open(os.path.join(d, "bitsandbytes", "functional.py"), "w") as f:
        f.write("def quantize_fp4(*a, **k):\n    raise NotImplementedError\ndef dequantize_fp4(*a, **k):\n    raise NotImplementedError\n")
    with open(os.path.join(d, "qMixtralLayer.py"), "w") as f:
        f.write("class QMixtralDecoderLayer:\n    pass\n")
    return d


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_impl(impl):
    """-> namespace(flow, eval, device, gptq_solver).  ``flow`` has the four modelutils_llama functions."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    if impl == "atom":
        from atom_amd.model import eval as E, modelutils_llama as F
        return types.SimpleNamespace(flow=F, eval=E, device="cuda:0")
    if impl == "atom_cpu":
        # OUR flow mirrors driving OUR classes on CPU (16-bit-activation configurations only: no kernel needed); the GPTQ solver is
        # the reference's module, found behind the drop-in directory on sys.path (build container only)
        if not torch.cuda.is_available():
            torch.cuda.synchronize = lambda *a, **k: None      # gptq.py:308 synchronises unconditionally; there is no GPU here
        sys.path[:0] = [os.path.join(ROOT, "atom_amd", "dropin"), _stub_dir()]
        if os.path.isdir(REF):
            sys.path.append(REF)
        F, E = importlib.import_module("modelutils_llama"), importlib.import_module("eval")
        assert F.__file__.startswith(ROOT) and E.__file__.startswith(ROOT)
        return types.SimpleNamespace(flow=F, eval=E, device="cpu")
    assert os.path.isdir(REF), "the reference tree is needed for impl=" + impl
    if not torch.cuda.is_available():
        torch.cuda.synchronize = lambda *a, **k: None          # gptq.py:308 synchronises unconditionally; there is no GPU here
    stubs = _stub_dir()
    if impl == "reference":
        sys.path[:0] = [stubs, REF]
        F, E = importlib.import_module("modelutils_llama"), importlib.import_module("eval")
        import quant as Qm
        assert Qm.__file__.startswith(REF) and F.__file__.startswith(REF)
    elif impl == "ref_flow_on_dropin":
        # OUR quant / qLinearLayer / qLlamaLayer under their top-level names; the reference's driver modules loaded from their files
        sys.path[:0] = [os.path.join(ROOT, "atom_amd", "dropin"), stubs]
        sys.path.append(REF)                                    # for gptq / tqdm-free helpers the driver imports by name
        import quant as Qm
        assert Qm.__file__.startswith(ROOT), Qm.__file__
        F = _load_by_path("ref_modelutils_llama", os.path.join(REF, "modelutils_llama.py"))
        E = _load_by_path("ref_eval", os.path.join(REF, "eval.py"))
        import atom_amd.model.qLlamaLayer as ours
        assert F.QLlamaDecoderLayer is ours.QLlamaDecoderLayer
    else:
        raise ValueError(impl)
    return types.SimpleNamespace(flow=F, eval=E, device="cpu")


class Recorder:
    """Forward hooks on every decoder layer (output rows of every sample) and on the QLinearLayers of sample 0 (their fake-quantised
    inputs and their outputs, every ROW_STEP-th token)."""

    def __init__(self, model):
        self.layer_out = [[] for _ in model.model.layers]
        self.lin = {}
        self.handles = []
        for i, layer in enumerate(model.model.layers):
            self.handles.append(layer.register_forward_hook(lambda _, inp, out, i=i: self.layer_out[i].append(out[0].detach().float().cpu())))
            for mod, proj in (("self_attn", "q_proj"), ("self_attn", "o_proj"), ("mlp", "gate_proj"), ("mlp", "down_proj")):
                lin = getattr(getattr(layer, mod), proj)
                key = f"L{i}.{proj}"
                self.handles.append(lin.register_forward_hook(lambda _, inp, out, key=key: self._lin(key, inp[0], out)))

    def _lin(self, key, x, y):
        if key + ".in" in self.lin:            # sample 0 only
            return
        self.lin[key + ".in"] = x.detach()[0, ::ROW_STEP].cpu().numpy().copy()
        self.lin[key + ".out"] = y.detach()[0, ::ROW_STEP].cpu().numpy().copy()

    def close(self):
        for h in self.handles:
            h.remove()


def weight_probe(model):
    """Bit-exact fingerprints of every projection's fp16 weight after the flow (a strided sample + the sum of magnitudes)."""
    out = {}
    for i, layer in enumerate(model.model.layers):
        for mod, proj in (("self_attn", "q_proj"), ("self_attn", "k_proj"), ("self_attn", "v_proj"), ("self_attn", "o_proj"),
                          ("mlp", "gate_proj"), ("mlp", "up_proj"), ("mlp", "down_proj")):
            w = getattr(getattr(layer, mod), proj).weight.detach().cpu()
            out[f"W{i}.{proj}.probe"] = w[::37, ::29].numpy().copy()
            out[f"W{i}.{proj}.abs_sum"] = np.float64(w.double().abs().sum().item())
    return out


def calibration_loader(ids, nsamples, seqlen):
    """What get_loaders returns as `dataloader`: a list of (inp [1, seqlen], tar) pairs (datautils.py:20-29)."""
    return [(ids[:, i * seqlen:(i + 1) * seqlen], None) for i in range(nsamples)]


def run(impl, config, tokens, gptq_q=None, device=None, prepare=None):
    """tokens: {"eval_<seqlen>": LongTensor [1, nsamples * seqlen], "calib": LongTensor [1, ...]}.  gptq_q: per-projection weights to
    install instead of running a solver (impl="atom" on the GPU box, where the reference's gptq.py does not exist).  prepare(model) is
    called between the quantisation passes and the evaluation."""
    from tests.flow_model import TinyLlamaForCausalLM, TokenStream, make_reorder_index, no_scales, paper_args
    mk, over, seqlens, use_gptq = CONFIGS[config]
    I = load_impl(impl)
    dev = device or I.device
    args = paper_args(**over)
    model = TinyLlamaForCausalLM(seqlen=seqlens[0], **mk).eval()
    F = I.flow
    model = F.reorder_model_llama(model, dev, args, make_reorder_index(model))
    if args.abits < 16:
        model = F.add_act_quant_wrapper_llama(model, dev, args, no_scales())
    if use_gptq:
        loader = calibration_loader(tokens["calib"], args.nsamples, model.seqlen)
        if gptq_q is not None:
            model = F.quantize_model_gptq_llama(model, dev, args, loader, solver=fixture_solver(gptq_q, model))
        else:
            model = F.quantize_model_gptq_llama(model, dev, args, loader)
    else:
        model = F.quantize_model_llama(model, dev, args)
    out = weight_probe(model)
    if prepare is not None:                                      # the GPU tests' calibration runs (tests/test_gpu_flow.py)
        prepare(model)
    if use_gptq and gptq_q is None:
        for i, layer in enumerate(model.model.layers):          # what the solver wrote (layer.weight.data = Q), for the GPU-side replay
            for name, lin in _named_linears(layer):
                out[f"Q{i}.{name}"] = lin.weight.detach().cpu().numpy().copy()
    for s in seqlens:
        model.seqlen = s
        rec = Recorder(model)
        ppl = I.eval.llama_eval(model, TokenStream(tokens[f"eval_{s}"]), dev)
        rec.close()
        out[f"ppl_{s}"] = np.float64(ppl)
        for i, rows in enumerate(rec.layer_out):
            x = torch.cat(rows, 0)                                # [nsamples, seqlen, hidden]
            out[f"s{s}.layer{i}.out"] = x[:, ::ROW_STEP].half().numpy().copy()
            out[f"s{s}.layer{i}.row_abs_sum"] = x.double().abs().sum(-1).numpy().copy()
        for k, v in rec.lin.items():
            out[f"s{s}.{k}"] = v
    return out


def _named_linears(layer):
    for mod, proj in (("self_attn", "q_proj"), ("self_attn", "k_proj"), ("self_attn", "v_proj"), ("self_attn", "o_proj"),
                      ("mlp", "gate_proj"), ("mlp", "up_proj"), ("mlp", "down_proj")):
        yield f"{mod}.{proj}", getattr(getattr(layer, mod), proj)


class fixture_solver:
    """A stand-in with the reference solver's three-call protocol whose ``fasterquant`` installs the weight the reference's GPTQ
    produced for this projection (stored by the golden generator), the way gptq.py:331 does: ``layer.weight.data = Q``.
    ``solver(lin, args)`` finds the projection by the identity of the QLinearLayer inside ``model``."""

    def __init__(self, gptq_q, model):
        self.q = gptq_q
        self.key = {}
        for i, layer in enumerate(model.model.layers):
            for name, lin in _named_linears(layer):
                self.key[id(lin)] = f"Q{i}.{name}"
        self.made = []

    def __call__(self, lin, args):
        key = self.key[id(lin)]
        q = torch.from_numpy(self.q[key])
        assert tuple(q.shape) == tuple(lin.weight.shape), (key, q.shape, lin.weight.shape)
        self.made.append(key)
        return _Replay(lin, q)


class _Replay:
    def __init__(self, lin, q):
        self.lin, self.q, self.batches, self.quantizer = lin, q, 0, None

    def add_batch(self, inp, out):
        assert inp.dim() == 3 and inp.shape[0] == 1 and out.shape[:2] == inp.shape[:2]          # what gptq.py:207-231 is written for
        self.batches += 1

    def fasterquant(self, percdamp, groupsize):
        assert self.batches > 0, "the forward hooks did not fire"
        w = self.lin.weight
        self.lin.weight.data = self.q.to(w.device).reshape(w.shape).to(w.data.dtype)          # gptq.py:331

    def free(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", required=True)
    ap.add_argument("--config", required=True)
    ap.add_argument("--tokens", required=True, help=".npz with eval_<seqlen> / calib token arrays")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    z = np.load(a.tokens)
    tokens = {k: torch.from_numpy(z[k].astype(np.int64))[None, :] for k in z.files}
    out = run(a.impl, a.config, tokens)
    np.savez_compressed(a.out, **out)
    print("FLOW_OK", a.impl, a.config, {k: float(v) for k, v in out.items() if k.startswith("ppl_")})


if __name__ == "__main__":
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    main()
