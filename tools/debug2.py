import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atom_amd import ops
from oracle import atom_oracle as O
from tests.helpers import *
z = np.load("tests/golden/c1_qlinear_16x512x512.npz")
outs = ops.reorder_fp16_i4(torch.from_numpy(z["x"]).cuda(), None, quant_mode="sim", clip=0.9, scale_layout="plain", return_dequant=True)
ref = O._quant_row_tail(z["x"], "sim", 0.9)
s4 = t2n(outs[3]).T
mm = np.argwhere(bits16(s4) != bits16(ref["s4"]))
for (r, g) in mm:
    blk = z["x"][r, g*128:(g+1)*128]
    am = np.abs(blk).max()
    print("s4 mism", r, g, "got", float(s4[r,g]), s4[r:r+1,g:g+1].view(np.uint16), "want", float(ref["s4"][r,g]), ref["s4"][r:r+1,g:g+1].view(np.uint16), "amax", float(am), "amax*0.9 f32", float(np.float32(am)*np.float32(0.9)))
for (N, K, sc) in [(256, 640, 0.05), (256, 768, 0.05), (64, 1152, 1.0)]:
    W = (np.random.default_rng(0).standard_normal((N, K)) * sc).astype(np.float16)
    b4, b8, sb, sb8, wq = ops.quant_weight_w4(torch.from_numpy(W).cuda(), 0.85, 2, return_fake_quant=True)
    w = O.quant_weight_sim(W, 0.85, 2)
    g4 = O.unpack_int4(t2n(b4))
    print(N, K, "weight: q4 mism", (g4 != w["q4"]).sum(), "q8 mism", (t2n(b8) != w["q8"]).sum(), "s4 mism", (bits16(t2n(sb)) != bits16(w["s4"])).sum(), "s8", (bits16(t2n(sb8)) != bits16(w["s8"])).sum())
    mm = np.argwhere(g4 != w["q4"])
    for (r, c) in mm[:6]:
        g = c // 128
        print("  ", r, c, "W", float(W[r, c]), "got", g4[r, c], "want", w["q4"][r, c], "scale got", float(t2n(sb)[g, r]), "want", float(w["s4"][g, r]), "ratio", float(W[r,c])/float(w["s4"][g,r]))
    mm = np.argwhere(bits16(t2n(sb)) != bits16(w["s4"]))
    for (g, n) in mm[:6]:
        blk = W[(n//2)*2:(n//2)*2+2, g*128:(g+1)*128]
        am = np.abs(blk).max()
        print("  scale", g, n, "got", float(t2n(sb)[g, n]), "want", float(w["s4"][g, n]), "amax", float(am), "amax*clip", float(np.float32(am)*np.float32(0.85)))
