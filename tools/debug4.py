import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
from tests.test_gpu_block import _build
from atom_amd.model.qLinearLayer import find_qlinear_layers
from atom_amd.model import quant
m, x, pos, mask, wsum = _build("cuda")
def hook(name):
    def f(mod, inp, out):
        xin = inp[0]
        ref = torch.nn.functional.linear(xin, mod.weight, mod.bias)
        codes = quant.get_codes(xin)
        d = (out.float() - ref.float()).abs().max().item()
        print(f"{name:22s} in {tuple(xin.shape)} codes={'y' if codes is not None else 'n'} N={mod.weight.shape[0]} K={mod.weight.shape[1]} max|hip-flinear|={d:.4f} ref_rms={ref.float().pow(2).mean().sqrt().item():.3f}")
        if codes is not None and d > 0.05:
            e = (out.float() - ref.float()).abs().reshape(-1, out.shape[-1])
            rows = (e.max(dim=1).values > 0.05).nonzero().flatten().tolist()
            cols = (e.max(dim=0).values > 0.05).nonzero().flatten().tolist()
            print("   bad rows", rows[:20], "n bad cols", len(cols), cols[:16])
    return f
for n_, l in find_qlinear_layers(m).items():
    l.register_forward_hook(hook(n_))
y = m(x.cuda(), attention_mask=mask.cuda(), position_ids=pos.cuda())[0]
