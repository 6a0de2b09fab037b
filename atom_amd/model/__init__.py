"""Drop-in replacements for the reference's ``model/quant.py``, ``model/qLinearLayer.py`` and ``model/qLlamaLayer.py``.

Put this directory FIRST on ``sys.path`` (or copy the three modules over the reference's) and the reference's
``modelutils_llama.py`` / ``gptq.py`` / ``outlier.py`` / ``main.py`` import them unchanged: same module names, class
names, function names and signatures.  With the paper configuration (W4A4, group 128, 128 INT8 keeper columns,
symmetric) every GEMM and every activation quantisation runs on the HIP kernels of libatom_hip.so.
"""
