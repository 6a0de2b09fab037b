"""Drop-in check at import level (build container only: needs /root/reference): the reference's own callers
(gptq.py, outlier.py, modelutils_llama.py) import `quant` / `qLinearLayer` / `qLlamaLayer` as top-level modules; with
atom_amd/dropin first on sys.path they get OUR classes.  No compute (CPU)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/model"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


def test_reference_callers_resolve_to_our_modules(tmp_path):
    # stubs for modules that are NOT ours and fail to import for unrelated reasons: bitsandbytes (not installed) and
    # the reference's qMixtralLayer (needs transformers 4.39's MixtralBlockSparseTop2MLP)
    (tmp_path / "bitsandbytes").mkdir()
    (tmp_path / "bitsandbytes" / "__init__.py").write_text("")
    (tmp_path / "bitsandbytes" / "functional.py").write_text("def quantize_fp4(*a, **k): raise NotImplementedError\n"
                                                             "def dequantize_fp4(*a, **k): raise NotImplementedError\n")
    (tmp_path / "qMixtralLayer.py").write_text("class QMixtralDecoderLayer: pass\n")
    code = textwrap.dedent(f"""
        import sys
        sys.path[:0] = [{str(os.path.join(ROOT, 'atom_amd', 'dropin'))!r}, {ROOT!r}, {str(tmp_path)!r}]
        sys.path.append({REF!r})
        import quant, qLinearLayer, qLlamaLayer
        assert quant.__file__.startswith({ROOT!r}) and qLlamaLayer.__file__.startswith({ROOT!r})
        import gptq, outlier
        import importlib.util                       # the reference's OWN modelutils_llama (dropin/ carries a mirror of that name too)
        spec = importlib.util.spec_from_file_location("ref_modelutils_llama", {REF!r} + "/modelutils_llama.py")
        modelutils_llama = importlib.util.module_from_spec(spec); spec.loader.exec_module(modelutils_llama)
        import modelutils_llama as mirror, eval as mirror_eval
        assert mirror.__file__.startswith({ROOT!r}) and mirror_eval.__file__.startswith({ROOT!r})
        for f in ("reorder_model_llama", "add_act_quant_wrapper_llama", "quantize_model_llama", "quantize_model_gptq_llama"):
            assert callable(getattr(mirror, f)) and callable(getattr(modelutils_llama, f))
        assert callable(mirror_eval.llama_eval)
        import atom_amd.model.qLinearLayer as ours
        assert gptq.QLinearLayer is ours.QLinearLayer and outlier.QLinearLayer is ours.QLinearLayer
        assert modelutils_llama.QLlamaDecoderLayer.__module__ == 'atom_amd.model.qLlamaLayer'
        assert modelutils_llama.find_qlinear_layers is ours.find_qlinear_layers
        from functools import partial
        import types, torch
        args = types.SimpleNamespace(abits=4, static=False)
        q = quant.Quantizer(args)
        q.configure(partial(modelutils_llama.quantize_activation_wrapper, args=args), None)
        assert q.act_quant.func is quant.quantize_activation_wrapper
        # the same exact-type registry walk the reference does (qLinearLayer.py:5-14) finds our layers
        lin = ours.QLinearLayer(torch.nn.Linear(256, 64, bias=False).half(), types.SimpleNamespace(wbits=4))
        box = torch.nn.Sequential(lin)
        assert list(modelutils_llama.find_qlinear_layers(box).values()) == [lin]
        print("DROPIN_OK")
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert "DROPIN_OK" in out.stdout, out.stderr[-3000:]


def test_dropin_eval_provides_every_name_the_reference_driver_takes(tmp_path):
    """main.py:4,12,211 -- ``from eval import *``, ``from eval import pattern_match``, ``eval_func = opt_eval`` -- with atom_amd/dropin
    ahead of the reference's model/ directory: the import must not fail, pattern_match must behave like the reference's, llama_eval is
    ours, and opt_eval (OPT: out of scope) resolves to the reference's own function behind us on the path."""
    (tmp_path / "tqdm.py").write_text("def tqdm(x, *a, **k):\n    return x\n")      # (the reference's eval.py imports it)
    code = textwrap.dedent(f"""
        import sys
        sys.path[:0] = [{str(os.path.join(ROOT, 'atom_amd', 'dropin'))!r}, {ROOT!r}, {str(tmp_path)!r}]
        sys.path.append({REF!r})
        from eval import *
        from eval import pattern_match
        import eval as E, importlib.util
        assert E.__file__.startswith({ROOT!r})
        names = ["arc_easy", "arc_challenge", "piqa", "boolq", "hellaswag"]
        spec = importlib.util.spec_from_file_location("ref_eval", {REF!r} + "/eval.py")
        ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
        for pats in (["arc_*"], ["piqa", "bool?"], ["*"], ["nothing"], ["arc_easy", "arc_*"]):
            assert sorted(pattern_match(pats, names)) == sorted(ref.pattern_match(pats, names)), pats
        assert llama_eval.__module__ == "atom_amd.model.eval"
        assert callable(opt_eval) and E._reference_eval().__file__.startswith({REF!r}) and callable(E._reference_eval().opt_eval)
        print("EVAL_OK")
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert "EVAL_OK" in out.stdout, out.stderr[-3000:]
