"""Golden fixture for the GPTQ bridge (atom_pack_weight_w4): run the UNMODIFIED reference GPTQ (model/gptq.py:197-331)
on one small QLinearLayer on CPU and record what it writes to ``layer.weight.data`` together with the FP32 scales it
used for every 128-column group (which the reference discards).  Build container only:

    python tests/golden/gen_golden_gptq.py          -> gptq_layer_64x512.npz

Pinned: GPTQ.add_batch / fasterquant (gptq.py:221-331), Quantizer_GPTQ.find_params (gptq.py:104-184), quantize_gptq
(gptq.py:27-61), the INT8 keeper (gptq.py:316-325 -> quant.py quantize_tensor).  Nothing of the reference is modified:
torch.cuda.synchronize (gptq.py:308, no GPU here) is stubbed and find_params is wrapped to log the scales.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import _import_reference, paper_args, n  # noqa: E402


def main():
    quant, qLinearLayer, qLlamaLayer = _import_reference()
    import gptq as G

    torch.cuda.synchronize = lambda *a, **k: None
    torch.manual_seed(1234)
    N, K, T = 64, 512, 640
    args = paper_args()
    lin = torch.nn.Linear(K, N, bias=False)
    with torch.no_grad():
        lin.weight.mul_(0.5)
        lin.weight[:, -128:] *= 4.0                              # keeper columns carry the outlier channels
    layer = qLinearLayer.QLinearLayer(lin.half(), args)
    w0 = layer.weight.data.clone()
    x = torch.randn(T, K)
    x[:, -128:] *= 10.0
    x = x.half()

    g = G.GPTQ(layer, n_out=args.keeper, keeper_precision=args.keeper_precision)
    g.quantizer = G.Quantizer_GPTQ()
    g.quantizer.configure(args.wbits, perchannel=True, sym=args.w_sym, mse=False, channel_group=args.weight_channel_group,
                          clip_ratio=args.w_clip_ratio, quant_type=args.quant_type)
    scales = []
    orig = g.quantizer.find_params

    def logged(xx, weight=False):
        orig(xx, weight=weight)
        scales.append(g.quantizer.scale.detach().clone().reshape(-1))

    g.quantizer.find_params = logged
    g.add_batch(x, None)
    g.fasterquant(percdamp=0.01, groupsize=args.weight_group_size)
    q = layer.weight.data
    assert q.dtype == torch.float16 and q.shape == (N, K)
    # find_params runs once on the whole INT4 part (gptq.py:244-245, overwritten) and then once per group (:285-287)
    assert len(scales) == 1 + (K - 128) // 128
    s32 = torch.stack(scales[1:], 0)                             # [G, N/2] FP32
    np.savez_compressed(os.path.join(HERE, "gptq_layer_64x512.npz"), w0=n(w0), q=n(q), s32=n(s32),
                        channel_group=np.int32(args.weight_channel_group))
    print("wrote gptq_layer_64x512.npz", "max|q-w0|", float((q.float() - w0.float()).abs().max()))


if __name__ == "__main__":
    main()
