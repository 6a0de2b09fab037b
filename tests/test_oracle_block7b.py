"""The oracle pinned at Llama-7B WIDTH against the unmodified reference (CPU, no GPU): tests/golden/llama_block_7b_wide.npz holds, for
256 token rows of the reference's QLlamaDecoderLayer.forward at hidden 4096 / intermediate 11008 (gen_golden_block7b.py), the
fake-quantised input of every projection (+ its recovered per-group scales) and 256 sampled output features of all seven
projections.  Here the oracle's own chain -- weight quantiser (quant_weight_sim <- qLinearLayer.py:42-78), integer view of the
reference's activations, the C restatement of the W4A4 GEMM contract -- must reproduce those outputs, and the oracle's
RMSNorm -> reorder -> quantise must reproduce the reference's first quantiser on all 256 rows.  The GPU tests then compare the HIP
path with this same fixture (tests/test_gpu_block.py) and with the oracle bit for bit (tests/test_gpu_gemm.py)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import atom_oracle as O  # noqa: E402
import c_oracle as C  # noqa: E402

PROJ = {   # projection -> (fixture key of its input, in-index, out-index, module path)
    "q": ("xq1", "qkv", None, ("self_attn", "q_proj")), "k": ("xq1", "qkv", None, ("self_attn", "k_proj")),
    "v": ("xq1", "qkv", None, ("self_attn", "v_proj")), "o": ("attn_q", "o", None, ("self_attn", "o_proj")),
    "gate": ("xq2", "gateup", "down", ("mlp", "gate_proj")), "up": ("xq2", "gateup", "down", ("mlp", "up_proj")),
    "down": ("act_q", "down", None, ("mlp", "down_proj")),
}


def act_codes(z, key):
    """Integer view of a fake-quantised reference tensor: (packed nibbles [M, K4/2], scales [G, M], keeper codes, keeper scales)."""
    v = z[key].astype(np.float32)
    s4, s8 = z["s4_" + key].astype(np.float32), z["s8_" + key].astype(np.float32)
    M, H = v.shape
    c4 = np.rint(v[:, :-128].reshape(M, -1, 128) / s4[..., None]).reshape(M, H - 128)
    c8 = np.rint(v[:, -128:] / s8[:, None])
    assert np.abs(c4).max() <= 8 and c4.max() <= 7 and np.abs(c8).max() <= 128 and c8.max() <= 127
    return O.pack_int4(c4.astype(np.int8)), np.ascontiguousarray(z["s4_" + key].T), c8.astype(np.int8), z["s8_" + key]


@pytest.fixture(scope="module")
def block():
    import torch
    import gen_golden_block7b as G7
    z = np.load(os.path.join(HERE, "golden", "llama_block_7b_wide.npz"))
    orig = G7.build_original()
    idx, x, _, _ = G7.make_inputs()
    return z, orig, {k: v.numpy() for k, v in idx.items()}, x[0].numpy(), torch


def test_recovered_scales_reproduce_the_reference_tensors(block):
    z = block[0]
    for key in ("xq1", "attn_q", "xq2", "act_q"):
        a4, sA, a8, sA8 = act_codes(z, key)
        c4 = O.unpack_int4(a4).astype(np.float32)
        M, K4 = c4.shape
        back = (c4.reshape(M, -1, 128) * z["s4_" + key].astype(np.float32)[..., None]).reshape(M, K4).astype(np.float16)
        assert np.array_equal(back, z[key][:, :-128]), key
        assert np.array_equal((a8.astype(np.float32) * z["s8_" + key].astype(np.float32)[:, None]).astype(np.float16), z[key][:, -128:]), key


def test_oracle_first_quantiser_matches_reference_on_256_rows(block):
    """input_layernorm -> reorder -> quantise (qLlamaLayer.py QLlamaRMSNorm + quant.py:187-231) on the 256 wide rows."""
    z, orig, idx, x, _ = block
    w = orig.input_layernorm.weight.detach().numpy().astype(np.float16)
    t = O.rmsnorm_reorder_quant(x[z["rows"]], w, 1e-5, idx["qkv"].astype(np.int64), mode="sim", clip=0.9)
    got = O.act_dequant_sim(t)
    differ = float((got != z["xq1"]).mean())
    print("oracle RMSNorm-quant vs the reference's xq1, 256 x 4096: fraction of differing elements", differ)
    assert differ <= 5e-3                        # the documented RMSNorm tolerance (one code step where it differs)


@pytest.mark.parametrize("name", list(PROJ))
def test_oracle_gemm_chain_matches_reference_outputs(block, name):
    """Oracle weight quantiser + the C restatement of the GEMM contract, fed the reference's own (integer-view) input, against the
    reference's output on 256 rows x 256 sampled features of each projection."""
    z, orig, idx, _, torch = block
    key, iin, iout, path = PROJ[name]
    W = getattr(getattr(orig, path[0]), path[1]).weight.detach()
    if iout is not None:
        W = W[torch.from_numpy(idx[iout])]
    W = W[:, torch.from_numpy(idx[iin])]
    cols = z["cols_" + name]
    need = np.unique(np.concatenate([cols, cols ^ 1]))                    # weight_channel_group = 2: a row's scale is its pair's
    q = O.quant_weight_sim(W[torch.from_numpy(need)].numpy().astype(np.float16), 0.85, 2)
    a4, sA, a8, sA8 = act_codes(z, key)
    D = C.gemm(a4, O.pack_int4(q["q4"]), sA, q["s4"], a8, q["q8"], sA8, q["s8"])
    got = D[:, np.searchsorted(need, cols)].astype(np.float64)
    ref = z["out_" + name].astype(np.float64)
    rel = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    rms = float(np.sqrt((ref ** 2).mean()))
    worst = float((np.abs(got - ref) / np.maximum(np.abs(ref), rms)).max())
    print(f"{name}: oracle vs reference on 256 x 256, relative Frobenius {rel:.2e}, worst element / max(|ref|, rms) {worst:.2e}")
    assert rel <= 1e-3 and worst <= 1e-2, (name, rel, worst)
