"""Top-level ``eval`` module for the reference's driver (main.py: ``from eval import ...``): with this directory ahead of
/path/to/Atom/model on PYTHONPATH the driver's flow functions are the MI355X-resident ones of atom_amd.model.eval (same names and
positional arguments).  Leave this file out of the path (load the reference's own eval by file) to drive OUR classes with the
reference's UNMODIFIED flow code instead -- both are tested (tests/test_flow_reference_cpu.py, tests/test_gpu_flow.py)."""
from atom_amd.model.eval import *  # noqa: F401,F403
from atom_amd.model import eval as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
