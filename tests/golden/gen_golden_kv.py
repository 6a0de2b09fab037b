"""Golden vectors for the KV-cache fake quantisation (reference model/quant.py:233-257 -> quantize_tensor(sym=False),
:143-145,173-181), produced by the UNMODIFIED reference on CPU.  Build container only:
    python tests/golden/gen_golden_kv.py      -> kv_fake_quant.npz"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import _import_reference, n  # noqa: E402


def main():
    quant, _, _ = _import_reference()
    g = torch.Generator().manual_seed(77)
    x = torch.randn(2, 3, 40, 128, generator=g)
    x[0, 0] *= 8.0                      # large keys
    x[0, 1] *= 1e-3                     # tiny values: scale near the 1e-5 clamp
    x[1, 2, :4] = 0.0                   # all-zero vectors: range clamps to 1e-5
    x[1, 1, :, ::2] = x[1, 1, :, ::2].abs()      # skewed: min far from -max
    x[1, 0, 0] = 3.25                   # constant vector
    x = x.half()
    out = {"x": n(x)}
    for bits, clip in ((4, 1.0), (4, 0.95), (8, 1.0)):
        args = types.SimpleNamespace(abits=bits, kv_clip_ratio=clip)
        out[f"k_b{bits}_c{clip}"] = n(quant.quantize_attn_k_wrapper(x.clone(), args))
        assert torch.equal(quant.quantize_attn_v_wrapper(x.clone(), args), torch.from_numpy(out[f"k_b{bits}_c{clip}"]))
    # the transposed (non-contiguous) view the attention layer actually passes (qLlamaLayer.py:238-249)
    xt = x.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    assert not xt.is_contiguous()
    args = types.SimpleNamespace(abits=4, kv_clip_ratio=1.0)
    assert torch.equal(quant.quantize_attn_k_wrapper(xt, args), torch.from_numpy(out["k_b4_c1.0"]))
    np.savez_compressed(os.path.join(HERE, "kv_fake_quant.npz"), **out)
    print("wrote kv_fake_quant.npz")


if __name__ == "__main__":
    main()
