"""Host-side logic of the operator surface that needs no GPU."""
import types

import numpy as np
import torch

from oracle import atom_oracle as O
from tests.helpers import bits16


def _args():
    return types.SimpleNamespace(abits=4, a_sym=True, act_group_size=128, keeper=128, keeper_precision=3, quant_type="int",
                                 exponential=False, a_clip_ratio=0.9, static=False, tiling=0)


def test_hot_config_is_bounded_by_the_kernels_hidden_limit():
    """The fused activation kernels take hidden <= 16384; wider MLP intermediates (Llama-30B/65B/70B: 17920 / 22016 / 28672,
    which the reference evaluates) must take the torch path instead of failing inside the library."""
    from atom_amd.model import quant as Q
    a = _args()
    assert Q.is_hot_act_config(a, 4096) and Q.is_hot_act_config(a, 11008) and Q.is_hot_act_config(a, 16384)
    for h in (17920, 22016, 28672):
        assert not Q.is_hot_act_config(a, h)


def test_wide_layer_activation_quant_matches_the_oracle_on_the_torch_path():
    """hidden 22016 -> quantize_activation_wrapper runs the reference's algorithm in torch ops (any device): same fake-quant
    tensor as the oracle's restatement of model/quant.py:187-231 (values equal; torch keeps the sign of a code that rounds to
    zero from below, -0.0, where the restatement writes +0.0)."""
    from atom_amd.model import quant as Q
    g = np.random.default_rng(5)
    x = g.standard_normal((3, 22016)).astype(np.float16)
    x[:, -128:] *= 25
    y = Q.quantize_activation_wrapper(torch.from_numpy(x), _args())
    want = O.act_dequant_sim(O.reorder_quant(x, np.arange(22016), "sim", 0.9))
    got = y.numpy()
    assert np.array_equal(got, want)
    nz = want != 0
    assert np.array_equal(bits16(got)[nz], bits16(want)[nz])
