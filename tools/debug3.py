import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
from tests.test_gpu_block import _build, _args
z = np.load("tests/golden/llama_block_512.npz")
m, x, pos, mask, wsum = _build("cuda")
y = m(x.cuda(), attention_mask=mask.cuda(), position_ids=pos.cuda())[0]
got = y.float().cpu().numpy().astype(np.float64); want = z["y"].astype(np.float64)
err = np.abs(got - want); rms = np.sqrt((want**2).mean())
print("rel fro", np.linalg.norm(got-want)/np.linalg.norm(want), "max", err.max(), "rms", rms, "quantiles", np.quantile(err, [0.5, 0.9, 0.99, 0.999]))
dx = want - z["x"].astype(np.float64); dg = got - z["x"].astype(np.float64)
print("block contribution rel fro", np.linalg.norm(dg-dx)/np.linalg.norm(dx), "rms contrib", np.sqrt((dx**2).mean()))
# same modules, F.linear path (no codes): drop packed
from atom_amd.model.qLinearLayer import find_qlinear_layers
for l in find_qlinear_layers(m).values(): l._packed = None
y2 = m(x.cuda(), attention_mask=mask.cuda(), position_ids=pos.cuda())[0]
g2 = y2.float().cpu().numpy().astype(np.float64)
print("F.linear-on-fake-quant path vs reference: rel fro", np.linalg.norm(g2-want)/np.linalg.norm(want), "max", np.abs(g2-want).max())
print("HIP path vs F.linear path: rel fro", np.linalg.norm(g2-got)/np.linalg.norm(want), "max", np.abs(g2-got).max())
