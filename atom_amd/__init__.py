"""atom_amd -- MI355X (gfx950) implementation of Atom's W4A4 mixed-precision GEMM hot path.

  atom_amd.ops      native ops with the reference's ``punica.ops`` surface (libatom_hip.so via the C ABI)
  atom_amd.model    drop-in ``quant`` / ``qLinearLayer`` / ``qLlamaLayer`` modules for the reference's model/
"""
__version__ = "0.1.0"
