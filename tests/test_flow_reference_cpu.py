"""FLOW-level drop-in, build container only (needs /root/reference): the reference's UNMODIFIED driver code --
``modelutils_llama.reorder_model_llama / add_act_quant_wrapper_llama / quantize_model_llama / quantize_model_gptq_llama``
(modelutils_llama.py:15-266), the reference's own GPTQ solver (gptq.py) and ``eval.llama_eval`` (eval.py:14-86), wired as
main.py:224-270 wires them -- is executed over a two-layer model twice: with the reference's classes and with OURS
(atom_amd/dropin first on sys.path), each in a fresh interpreter (tests/flow_run.py).

On CPU only the configurations that need no kernel can run through our classes (W4, 16-bit activations: the W4A4 hot path has no
CPU fallback by design -- tests/test_gpu_flow.py covers it on the GPU against goldens written by the reference flow).  (A third run puts this repository's flow mirrors -- the code the GPU tests execute -- in place of the reference's flow code.)  For those
the runs must agree BIT FOR BIT on everything the flow produces: every projection's weight after reorder + RTN / GPTQ
(``layer.weight.data = Q``, hooks, per-layer .cpu()/.to(dev) round trips, layer replacement inside model.model.layers), the
fake-quantised K/V path, every layer's output on every sample, and the perplexity."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/model"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


def _tokens(tmp_path, golden_dir, prefix):
    z = np.load(os.path.join(golden_dir, "flow_tokens.npz"))
    p = str(tmp_path / f"tokens_{prefix}.npz")
    np.savez(p, **{k.split(".", 1)[1]: z[k] for k in z.files if k.startswith(prefix + ".")})
    return p


def _run(impl, config, tokens, out):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "flow_run.py"), "--impl", impl, "--config", config,
                        "--tokens", tokens, "--out", out], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "FLOW_OK" in r.stdout, r.stderr[-3000:]
    return np.load(out)


@pytest.mark.parametrize("config,stream", [("rtn_w4a16", "rtn_w4a4"), ("gptq_w4a16", "gptq_w4a4"), ("rtn_w4a16_gqa", "rtn_w4a4_gqa")])
def test_unmodified_reference_flow_drives_our_classes_bit_for_bit(tmp_path, golden_dir, config, stream):
    tok = _tokens(tmp_path, golden_dir, stream)
    ref = _run("reference", config, tok, str(tmp_path / "ref.npz"))
    ours = _run("ref_flow_on_dropin", config, tok, str(tmp_path / "ours.npz"))
    # ... and the flow code the GPU tests run: this repository's mirrors (atom_amd.model.modelutils_llama / eval) over our classes
    mirror = _run("atom_cpu", config, tok, str(tmp_path / "mirror.npz"))
    assert set(ref.files) == set(ours.files) == set(mirror.files) and len(ref.files) >= 49
    for k in ref.files:
        assert np.array_equal(ref[k], ours[k]), k
        assert np.array_equal(ref[k], mirror[k]), ("mirror", k)
    if config.startswith("gptq"):
        assert any(k.startswith("Q1.") for k in ref.files)          # the solver's weights of the SECOND layer: its Hessians came from
        #                                                             layer 0's outputs computed by our classes


def test_flow_goldens_are_what_the_generator_writes(golden_dir):
    """The committed W4A4 goldens carry the keys the GPU test reads, for both layers and both sequence lengths."""
    z = np.load(os.path.join(golden_dir, "flow_rtn_w4a4.npz"))
    for s in (96, 320):
        assert f"ppl_{s}" in z.files and z[f"s{s}.layer1.out"].shape == (4, s // 8, 512)
        assert z[f"s{s}.L1.down_proj.in"].shape == (s // 8, 1408)
    g = np.load(os.path.join(golden_dir, "flow_gptq_w4a4.npz"))
    assert g["Q0.mlp.down_proj"].shape == (256, 384) and g["Q1.self_attn.q_proj"].dtype == np.float16
