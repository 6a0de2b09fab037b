"""Top-level ``qLinearLayer`` module for the reference's ``from qLinearLayer import ...`` statements: put this directory (and the repo root)
on PYTHONPATH ahead of /path/to/Atom/model.  Everything is re-exported from atom_amd.model.qLinearLayer (same class objects,
so ``type(m) == QLinearLayer`` checks keep working)."""
from atom_amd.model.qLinearLayer import *  # noqa: F401,F403
from atom_amd.model import qLinearLayer as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
