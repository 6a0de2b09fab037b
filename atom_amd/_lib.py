"""ctypes binding of libatom_hip.so (the C ABI declared in include/atom_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, an exception is
raised.  (The CPU oracle lives in /oracle and is test infrastructure only.)
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libatom_hip.so")
CSRC = os.path.join(_HERE, "csrc")

OK = 0
ERR_INVALID_ARG, ERR_SHAPE, ERR_ALIGN, ERR_LAUNCH = -22, -33, -14, -5   # ATOM_ERR_*
SCALE_LAYOUT_REF = 0
SCALE_LAYOUT_PLAIN = 1
QUANT_KERNEL = 0
QUANT_SIM = 1
QUANT_WIDE_CODES = 0x100   # ATOM_QUANT_WIDE_CODES
A_WIDE = 0x100             # ATOM_A_WIDE
QUANT_F6_CODES = 0x200     # ATOM_QUANT_F6_CODES
AB_F6 = 0x200              # ATOM_AB_F6
O4_REF_EXTREMA = 0x800     # ATOM_O4_REF_EXTREMA
B_F6S = 0x400              # ATOM_B_F6S: float32 weight scales appended to the F6 weight buffer
WS_WEIGHT_CACHED = 0x1000  # ATOM_WS_WEIGHT_CACHED: the workspace already holds this weight's F6 form
B_SCALE_PAIRS = 0x2000     # ATOM_B_SCALE_PAIRS: output channels 2j, 2j+1 share their weight scales (weight_channel_group = 2)
WS_VERIFY = 0x4000         # ATOM_WS_VERIFY: debug call -- the two assertions above are checked on the device first (synchronises)
Q_REORDER, Q_RMSNORM, Q_ADD_RMSNORM, Q_SILU_MUL = 1, 2, 3, 4     # atom_gemm_w4a4_multi_q: q_op
F6_PITCH = 104

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_f32 = ctypes.c_float

# name -> (restype, argtypes); must list every symbol of include/atom_hip.h
SIGNATURES = {
    "atom_version": (ctypes.c_char_p, []),
    "atom_strerror": (ctypes.c_char_p, [_int]),
    "atom_scale_size": (ctypes.c_size_t, [_i64, _int]),
    "atom_gemm_w4a4_f16": (_int, [_vp] * 9 + [_i64, _i64, _i64, _int, _int, _int, _vp]),
    "atom_gemm_w4a4_workspace_bytes": (ctypes.c_size_t, [_i64, _i64, _i64]),
    "atom_gemm_w4a4_ws_recodes": (_int, [_i64, _i64, _i64]),
    "atom_gemm_w4a4_ws_recodes_cached": (_int, [_i64, _i64, _i64]),
    "atom_gemm_w4a4_packed_order": (_int, [_i64, _i64, _i64, _int]),
    "atom_check_scale_pairs": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "atom_verify_weight_f6s": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "atom_gemm_w4a4_f16_ws": (_int, [_vp] * 9 + [_i64, _i64, _i64, _int, _int, _int, _vp, ctypes.c_size_t, _vp]),
    "atom_gemm_w4a4_multi_fits": (_int, [_i64, _i64, _int, _i64]),
    "atom_gemm_w4a4_multi": (_int, [_vp] * 11 + [ctypes.c_uint, _vp, _i64, _i64, _int, _i64, _int, _int, _int, _vp]),
    "atom_gemm_w4a4_multi_q_fits": (_int, [_int, _i64, _i64, _int, _i64]),
    "atom_gemm_w4a4_multi_q": (_int, [_int] + [_vp] * 5 + [_f32, _f32] + [_vp] * 7 + [ctypes.c_uint, _vp, _i64, _i64, _int, _i64, _int, _int, _vp]),
    "atom_gemm_w4a4_o4": (_int, [_vp] * 10 + [_i64, _i64, _i64, _int, _int, _int, _vp]),
    "atom_gemm_w4a4_o4_workspace_bytes": (ctypes.c_size_t, [_i64, _i64, _i64]),
    "atom_gemm_w4a4_o4_ws": (_int, [_vp] * 10 + [_i64, _i64, _i64, _int, _int, _int, _vp, ctypes.c_size_t, _vp]),
    "atom_reorder_quant_f16": (_int, [_vp, _vp, _i64, _int, _int, _f32, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "atom_rmsnorm_reorder_quant_f16": (_int, [_vp, _vp, _f32, _vp, _i64, _int, _int, _f32, _int,
                                              _vp, _vp, _vp, _vp, _vp, _vp]),
    "atom_add_rmsnorm_reorder_quant_f16": (_int, [_vp, _vp, _vp, _vp, _f32, _vp, _i64, _int, _int, _f32, _int,
                                                  _vp, _vp, _vp, _vp, _vp, _vp]),
    "atom_silu_mul_quant_f16": (_int, [_vp, _vp, _i64, _int, _int, _f32, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "atom_quant_weight_w4": (_int, [_vp, _i64, _i64, _f32, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "atom_pack_weight_w4": (_int, [_vp, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "atom_f6_rows": (ctypes.c_size_t, [_i64]),
    "atom_repack_weight_f6": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "atom_repack_act_f6": (_int, [_vp, _vp, _i64, _i64, _int, _vp, _vp]),
    "atom_f6_weight_bytes": (ctypes.c_size_t, [_i64, _i64]),
    "atom_gemm_w4a4_f6_order": (ctypes.c_int, [_i64, _i64, _i64]),
    "atom_repack_weight_f6s": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "atom_kv_fake_quant_f16": (_int, [_vp, _vp, _i64, _int, _i64, _i64, _i64, _i64, _int, _f32, _vp]),
    "atom_kv_append_i4": (_int, [_vp] * 10 + [_i64] + [_int] * 6 + [_vp]),
    "atom_kv_quant_append_f32": (_int, [_vp] * 7 + [_int] * 6 + [_vp]),
    "atom_gemm_w4a4_f32": (_int, [_vp] * 9 + [_i64, _i64, _i64, _int, _int, _int, _vp]),
    "atom_gemm_w4a4_silu_mul_quant_f6": (_int, [_vp] * 6 + [_i64, _i64, _i64, _int, _int, _int, _f32, _int] + [_vp] * 6),
    "atom_batch_decode_i4_workspace_bytes": (ctypes.c_size_t, [_int, _int, _int, _int]),
    "atom_batch_decode_i4_splits": (_int, [_int, _int, _int, _int]),
    "atom_gemm_w4a4_multi_merge_q_fits": (_int, [_i64, _i64, _int, _i64, _int]),
    "atom_gemm_w4a4_multi_merge_q": (_int, [_vp, _int, _vp, _f32] + [_vp] * 7 + [ctypes.c_uint, _vp, _i64, _i64, _int, _i64, _int, _int, _vp]),
    "atom_batch_decode_i4": (_int, [_vp] * 7 + [_int] * 6 + [_f32, _f32, _int, _vp, ctypes.c_size_t, _vp]),
    "atom_batch_decode_append_i4": (_int, [_vp] * 9 + [_int] * 6 + [_f32, _f32, _int, _vp, ctypes.c_size_t, _vp]),
}

_lib = None


class AtomHipError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into atom_amd/libatom_hip.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "all"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
        print(r.stderr)
    if r.returncode != 0:
        raise AtomHipError("building libatom_hip.so failed:\n" + r.stderr)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AtomHipError(
                f"{LIB_PATH} not found: build it with `make -C atom_amd/csrc` (or __graft_entry__.build()). "
                "There is no CPU fallback for the Atom W4A4 path.")
        # torch first: PyTorch-ROCm ships its own libamdhip64, and the process must have ONE HIP runtime -- dlopen'ed before torch, this
        # library binds the system's copy, torch then loads its own, and every launch on one of torch's streams fails (HIP launch
        # failed) because the kernels are registered with the other runtime (seen with build() + smoke() in one process, round 4)
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)           # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int, what: str) -> None:
    if status != OK:
        msg = lib().atom_strerror(status).decode()
        raise AtomHipError(f"{what} failed: {msg} (status {status})")


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


def current_stream(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
