import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atom_amd import ops
from oracle import atom_oracle as O
from tests.helpers import *

z = np.load("tests/golden/c1_qlinear_16x512x512.npz")
# --- act quant sim
outs = ops.reorder_fp16_i4(torch.from_numpy(z["x"]).cuda(), None, quant_mode="sim", clip=0.9, scale_layout="plain", return_dequant=True)
ref = O._quant_row_tail(z["x"], "sim", 0.9)
q4 = O.unpack_int4(t2n(outs[1]).view(np.uint8)); q8 = t2n(outs[0])
s4 = t2n(outs[3]).T; s8 = t2n(outs[2])
print("act: q4 mism", (q4 != ref["q4"]).sum(), "q8 mism", (q8 != ref["q8"]).sum(), "s4 mism", (bits16(s4) != bits16(ref["s4"])).sum(), "s8 mism", (bits16(s8) != bits16(ref["s8"])).sum())
xq = t2n(outs[4]); xr = O.act_dequant_sim(ref)
mm = np.argwhere(bits16(xq) != bits16(xr))
print("xq mism", len(mm), mm[:10])
for (r, c) in mm[:10]:
    g = c // 128
    print(r, c, "x", float(z["x"][r, c]), "got", float(xq[r, c]), "want", float(xr[r, c]), "scale got", float(s4[r, g]) if g < 3 else float(s8[r]), "want", float(ref["s4"][r, g]) if g < 3 else float(ref["s8"][r]))
# --- weight quant
W = z["W"]
b4, b8, sb, sb8, wq = ops.quant_weight_w4(torch.from_numpy(W).cuda(), 0.85, 2, return_fake_quant=True)
w = O.quant_weight_sim(W, 0.85, 2)
g4 = O.unpack_int4(t2n(b4))
print("weight: q4 mism", (g4 != w["q4"]).sum(), "q8 mism", (t2n(b8) != w["q8"]).sum(), "s4 mism", (bits16(t2n(sb)) != bits16(w["s4"])).sum(), "s8", (bits16(t2n(sb8)) != bits16(w["s8"])).sum())
mm = np.argwhere(g4 != w["q4"])
for (r, c) in mm[:10]:
    g = c // 128
    print(r, c, "W", float(W[r, c]), "got", g4[r, c], "want", w["q4"][r, c], "scale got", float(t2n(sb)[g, r]), "want", float(w["s4"][g, r]), "ratio", float(W[r,c])/float(w["s4"][g,r]))
mm = np.argwhere(bits16(t2n(sb)) != bits16(w["s4"]))
for (g, n) in mm[:10]:
    blk = W[(n//2)*2:(n//2)*2+2, g*128:(g+1)*128]
    print("scale", g, n, "got", float(t2n(sb)[g, n]), "want", float(w["s4"][g, n]), "amax", float(np.abs(blk).max()))

# --- GEMM doubling / determinism
M, N, K = 4096, 4096, 4096
d = rand_gemm_operands(M, N, K, seed=11)
a = to_device(d, "plain")
base = ops.dense_layer_gemm_i4_fp16(*a, scale_layout="plain")
base2 = ops.dense_layer_gemm_i4_fp16(*a, scale_layout="plain")
print("deterministic:", torch.equal(base, base2), (base != base2).sum().item())
a2 = list(a); a2[2] = a[2] * 2; a2[6] = a[6] * 2
dbl = ops.dense_layer_gemm_i4_fp16(*a2, scale_layout="plain")
ne = (dbl != base * 2)
print("doubling mismatches:", ne.sum().item(), "nan", torch.isnan(base).sum().item(), "inf", torch.isinf(base * 2).sum().item())
idx = ne.nonzero()[:10]
for (r, c) in idx.tolist():
    print(r, c, base[r, c].item(), dbl[r, c].item())
a3 = list(a); a3[2] = a[2] * 0.5; a3[6] = a[6] * 0.5
hlf = ops.dense_layer_gemm_i4_fp16(*a3, scale_layout="plain")
print("halving mismatches:", (hlf != base * 0.5).sum().item())
