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


def _load_reference_quant():
    """the unmodified reference model/quant.py (build container only), with a stub for its bitsandbytes import"""
    import importlib.util, os, sys, types as T
    path = "/root/reference/model/quant.py"
    if not os.path.exists(path):
        return None
    bnb = T.ModuleType("bitsandbytes"); fn = T.ModuleType("bitsandbytes.functional")
    fn.quantize_fp4 = fn.dequantize_fp4 = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    bnb.functional = fn
    sys.modules.setdefault("bitsandbytes", bnb); sys.modules.setdefault("bitsandbytes.functional", fn)
    spec = importlib.util.spec_from_file_location("_ref_quant", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_non_hot_branches_of_the_operator_surface():
    """quantize_tensor(exponential=True), the static Quantizer branch and quant_type="fp": behaviour, not stubs.  The first two are
    compared with the unmodified reference module where it is present (bit for bit, fp32 and fp16); FP4 is checked against its
    definition (bitsandbytes is not installed: unpinned)."""
    from atom_amd.model import quant as Q
    ref = _load_reference_quant()
    g = torch.Generator().manual_seed(0)
    for dt in (torch.float32, torch.float16):
        w = (torch.randn(6, 256, generator=g) * 3).to(dt)
        for sym in (True, False):
            ours = Q.quantize_tensor(w.clone(), 4, 128, 0, sym, exponential=True)
            assert ours.shape == w.shape and torch.isfinite(ours.float()).all()
            if ref is not None:
                assert torch.equal(ours, ref.quantize_tensor(w.clone(), 4, 128, 0, sym, exponential=True))
    # levels: sym exponent format of 4 bits = +-scales * 2^e, e = 0..7
    w = torch.tensor([[128.0, 64.0, 50.0, 1.0, -3.0, 0.2, -100.0, 20.0]])
    out = Q.quantize_tensor(w, 4, 8, 0, True, exponential=True)
    assert torch.equal(out, torch.tensor([[128.0, 64.0, 64.0, 1.0, -2.0, 1.0, -128.0, 16.0]]))
    # FP4: every output is absmax * (+-) one of the 8 magnitudes; the absmax element is preserved
    w = torch.randn(4, 64, generator=g)
    out = Q.quantize_tensor(w, 4, 0, 0, True, quant_type="fp")
    lv = torch.tensor(Q._FP4_LEVELS) / 12.0
    ratio = (out / w.abs().amax(dim=-1, keepdim=True)).abs()
    assert ((ratio.unsqueeze(-1) - lv).abs().min(dim=-1).values < 1e-6).all()
    assert torch.equal(out.abs().amax(dim=-1), w.abs().amax(dim=-1))
    # static Quantizer
    args = _args(); args.static = True; args.keeper = 128; args.act_group_size = 128
    x = torch.randn(3, 384, generator=g)
    scales = torch.rand(6, 1, generator=g) * 0.3 + 0.05
    q = Q.Quantizer(args); q.configure(None, scales.clone())
    got = q(x.clone())
    if ref is not None:
        r = ref.Quantizer(args); r.configure(None, scales.clone())
        assert torch.equal(got, r(x.clone()))
    assert torch.equal(got[:, :128], x[:, :128]) and not torch.equal(got[:, 128:], x[:, 128:])
