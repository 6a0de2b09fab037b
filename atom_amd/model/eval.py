"""HIP-path mirror of the reference's ``model/eval.py:14-86`` (``llama_eval``): perplexity of a (quantised) Llama over a token
stream, layer by layer.

Same arithmetic per sample as the reference -- every layer is called on ``inps[j].unsqueeze(0)`` with the attention mask / position
ids captured from the model's own forward, the loss is ``CrossEntropyLoss`` over the shifted logits, the result
``exp(sum(nll) / (nsamples * seqlen))`` -- so the number is comparable to the reference's.  What differs is residency: the layers
stay on ``dev`` (an MI355X holds the model; ``offload=True`` restores the reference's per-layer GPU <-> CPU bounce), and
``rows_per_call`` > 1 stacks that many samples into one layer call (tokens are quantised per token and GEMM rows are independent, so
this only changes which tile kernel -- and hence which fixed K order -- the W4A4 GEMMs take; default 1 = the reference's calls).
``return_details=True`` also returns the per-sample NLLs and the final hidden states (what the flow tests compare).
"""
from __future__ import annotations

import fnmatch

import torch
from torch import nn

from .modelutils_llama import capture_first_layer_inputs


def pattern_match(patterns, source_list):
    """The names of ``source_list`` that match any of the shell-style ``patterns`` (reference eval.py:6-11; main.py:314 picks the
    lm-eval tasks with it).  Unique, in the order of ``source_list`` (the reference returns them in set order)."""
    return [name for name in source_list if any(fnmatch.fnmatchcase(name, pat) for pat in patterns)]


def _for_batch(kw, n):
    """The captured layer keyword arguments ([1, ...] mask / position ids / (cos, sin)) broadcast to n samples."""
    wide = lambda t: t.expand(n, *t.shape[1:])
    return {k: None if v is None else wide(v) if torch.is_tensor(v) else tuple(wide(t) for t in v) for k, v in kw.items()}


@torch.no_grad()
def llama_eval(model, testenc, dev, offload: bool = False, rows_per_call: int = 1, return_details: bool = False):
    testenc = testenc.input_ids
    seqlen = model.seqlen
    nsamples = testenc.numel() // seqlen
    layers = model.model.layers
    batches = (testenc[:, i * seqlen:(i + 1) * seqlen] for i in range(nsamples))
    inps, kw = capture_first_layer_inputs(model, batches, nsamples, dev, offload)
    outs = torch.zeros_like(inps)
    step = max(1, int(rows_per_call))
    for i in range(len(layers)):
        layer = layers[i].to(dev)
        for j in range(0, nsamples, step):
            n = min(step, nsamples - j)
            o = layer(inps[j:j + n], **(kw if n == 1 else _for_batch(kw, n)))
            o = o[0] if isinstance(o, (tuple, list)) else o            # (decoder layers of newer transformers return the bare tensor)
            assert o.shape[0] == n, f"layer {i} returned {tuple(o.shape)} for {n} samples"
            outs[j:j + n] = o
        layers[i] = layer.cpu() if offload else layer
        inps, outs = outs, inps
    head = model.lm_head.to(dev)
    final_norm = model.model.norm.to(dev) if model.model.norm is not None else None
    model.lm_head = head
    if final_norm is not None:
        model.model.norm = final_norm
    targets = testenc.to(dev)[0, :nsamples * seqlen].view(nsamples, seqlen)

    def sample_nll(i):
        """seqlen x the mean token loss of sample i, the loss taken on the model's own (fp16) logits as the reference takes it
        (eval.py:68-79: the mean over seqlen - 1 predictions, scaled by seqlen)."""
        h = inps[i:i + 1]
        logits = head(h if final_norm is None else final_norm(h))[0, :-1]
        return nn.functional.cross_entropy(logits, targets[i, 1:]).float() * seqlen

    nlls = torch.stack([sample_nll(i) for i in range(nsamples)])
    ppl = torch.exp(nlls.sum() / (nsamples * seqlen)).item()
    if return_details:
        return ppl, nlls.cpu(), inps
    return ppl
