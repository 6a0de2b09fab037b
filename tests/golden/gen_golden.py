"""Generate golden vectors by running the UNMODIFIED reference modules (read-only import from
/root/reference/model).  Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/gen_golden.py

Outputs small .npz fixtures next to this file.  Everything is seeded; re-running reproduces the
files bit for bit (torch 2.10 CPU kernels).

What is pinned (reference file:line):
  * quantize_tensor                 model/quant.py:118-183
  * quantize_tensor_channel_group   model/quant.py:68-107
  * QLinearLayer.quant / .forward   model/qLinearLayer.py:32-35,42-78
  * quantize_activation_wrapper     model/quant.py:187-231
  * QLlamaRMSNorm.forward           model/qLlamaLayer.py:141-151 (HF LlamaRMSNorm + index_select + quant)
  * QLlamaMLP's act_fn(gate)*up -> act_quant   model/qLlamaLayer.py:345-351
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/model"


def _import_reference():
    # bitsandbytes is imported by quant.py:4 only for --quant_type fp; stub the two names.
    stub_root = tempfile.mkdtemp()
    os.makedirs(os.path.join(stub_root, "bitsandbytes"))
    open(os.path.join(stub_root, "bitsandbytes", "__init__.py"), "w").close()
    with open(os.path.join(stub_root, "bitsandbytes", "functional.py"), "w") as f:
        f.write("def quantize_fp4(*a, **k):\n    raise NotImplementedError\n"
                "def dequantize_fp4(*a, **k):\n    raise NotImplementedError\n")
    sys.path.insert(0, REF)
    sys.path.insert(0, stub_root)
    import quant  # noqa
    import qLinearLayer  # noqa
    import qLlamaLayer  # noqa
    return quant, qLinearLayer, qLlamaLayer


def paper_args(**over):
    """scripts/run_atom_ppl.sh:11-15 (W4A4 sym, g128, channel_group 2, clips 0.9/0.85, keeper 128 INT8)."""
    d = dict(wbits=4, abits=4, a_sym=True, w_sym=True, act_group_size=128, weight_group_size=128,
             weight_channel_group=2, keeper=128, keeper_precision=3, a_clip_ratio=0.9,
             w_clip_ratio=0.85, kv_clip_ratio=1.0, tiling=0, exponential=False, quant_type="int",
             static=False, reorder=True)
    d.update(over)
    return types.SimpleNamespace(**d)


def n(t):
    return t.detach().cpu().numpy()


def main():
    quant, qLinearLayer, qLlamaLayer = _import_reference()
    from transformers.models.llama.modeling_llama import LlamaRMSNorm
    torch.set_num_threads(1)
    args = paper_args()

    # ---- 1. quantize_tensor known-answer vectors incl. edge cases -------------------------------
    g = torch.Generator().manual_seed(1234)
    rows = []
    rows.append(torch.randn(8, 128, generator=g))                       # ordinary
    rows.append(torch.randn(4, 128, generator=g) * 30.0)                # large
    rows.append(torch.randn(4, 128, generator=g) * 1e-3)                # small
    rows.append(torch.zeros(2, 128))                                    # all-zero group (amax clamp)
    rows.append(torch.full((1, 128), 3e-6))                             # below the 1e-5 clamp
    rows.append(torch.full((1, 128), 1e-5))                             # at the clamp
    z = torch.zeros(2, 128); z[0, 5] = 7.0; z[1, 9] = -65504.0          # single spike / fp16 max
    rows.append(z)
    t = torch.arange(-64, 64).float().reshape(1, 128) * 0.5             # exact .5 ties for round-half-even
    rows.append(t)
    v = torch.cat(rows, 0).half()
    kat = {"v": n(v)}
    for bits, clip in [(4, 0.9), (4, 0.85), (4, 1.0), (8, 1.0)]:
        out = quant.quantize_tensor(v.clone(), n_bits=bits, group_size=0, tiling=0, sym=True,
                                    clip_ratio=clip, exponential=False)
        kat[f"out_b{bits}_c{clip}"] = n(out)
    np.savez_compressed(os.path.join(HERE, "kat_quantize_tensor.npz"), **kat)

    # ---- 2. config 1: QLinearLayer M=16 N=512 K=512 ---------------------------------------------
    torch.manual_seed(0)
    lin = torch.nn.Linear(512, 512, bias=False).half()
    # give the last 128 input channels outlier-sized weights, like a reordered layer
    W0 = lin.weight.data.clone()
    ql = qLinearLayer.QLinearLayer(lin, args)
    ql.quant()
    x = torch.randn(16, 512, generator=torch.Generator().manual_seed(1)).half()
    x[:, -128:] *= 20.0
    xq = quant.quantize_activation_wrapper(x.clone(), args)
    y = ql(xq)
    np.savez_compressed(os.path.join(HERE, "c1_qlinear_16x512x512.npz"),
                        W=n(W0), Wq=n(ql.weight), x=n(x), xq=n(xq), y=n(y))

    # ---- 3. reorder + act quant (QLlamaAttention o_proj input path) ------------------------------
    gen = torch.Generator().manual_seed(2)
    xr = torch.randn(9, 1024, generator=gen).half()
    xr[:, torch.randperm(1024, generator=gen)[:128]] *= 25.0
    idx = torch.randperm(1024, generator=gen)
    sel = torch.index_select(xr, 1, idx)
    xrq = quant.quantize_activation_wrapper(sel.clone(), args)
    np.savez_compressed(os.path.join(HERE, "reorder_quant_9x1024.npz"),
                        x=n(xr), idx=n(idx).astype(np.int16), xq=n(xrq))

    # ---- 4. QLlamaRMSNorm: HF RMSNorm -> index_select -> act_quant -------------------------------
    H = 1024
    norm = LlamaRMSNorm(H, eps=1e-5)
    norm.weight.data = (1.0 + 0.1 * torch.randn(H, generator=gen))
    norm = norm.half()
    qn = qLlamaLayer.QLlamaRMSNorm(norm, args)
    qn.register_buffer("reorder_index", idx.clone())
    from functools import partial
    qn.act_quant.configure(partial(quant.quantize_activation_wrapper, args=args), None)
    xn = (torch.randn(7, H, generator=gen) * 3.0).half()
    xn[:, 17] *= 30.0
    normed = norm(xn)
    out = qn(xn)
    np.savez_compressed(os.path.join(HERE, "rmsnorm_quant_7x1024.npz"),
                        x=n(xn), w=n(norm.weight), eps=np.float32(1e-5), idx=n(idx).astype(np.int16),
                        normed=n(normed), xq=n(out))

    # ---- 5. SiLU(gate)*up -> act_quant -----------------------------------------------------------
    a = (torch.randn(5, 1408, generator=gen) * 2.0).half()
    b = (torch.randn(5, 1408, generator=gen) * 2.0).half()
    prod = torch.nn.functional.silu(a) * b
    pq = quant.quantize_activation_wrapper(prod.clone(), args)
    np.savez_compressed(os.path.join(HERE, "silu_mul_quant_5x1408.npz"),
                        a=n(a), b=n(b), prod=n(prod), xq=n(pq))

    # ---- 6. weight quant alone, odd-ish shape (N=256, K=640) --------------------------------------
    Wt = (torch.randn(256, 640, generator=gen) * 0.05).half()
    Wt[:, -128:] *= 8.0
    lin2 = torch.nn.Linear(640, 256, bias=False).half()
    lin2.weight.data = Wt.clone()
    ql2 = qLinearLayer.QLinearLayer(lin2, args)
    ql2.quant()
    np.savez_compressed(os.path.join(HERE, "weight_quant_256x640.npz"), W=n(Wt), Wq=n(ql2.weight))

    print("golden fixtures written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f"  {f}: {os.path.getsize(os.path.join(HERE, f))} bytes")


if __name__ == "__main__":
    main()
