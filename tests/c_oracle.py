"""ctypes view of oracle/libatom_oracle.so (the plain-C restatement).  Checker side only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "libatom_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = ctypes.CDLL(_SO)
        L.oracle_scale_index.restype = ctypes.c_int
        L.oracle_scale_size.restype = ctypes.c_long
        L.oracle_scale_size.argtypes = [ctypes.c_long]
        L.oracle_f2h.restype = ctypes.c_uint16
        L.oracle_f2h.argtypes = [ctypes.c_float]
        L.oracle_h2f.restype = ctypes.c_float
        L.oracle_h2f.argtypes = [ctypes.c_uint16]
        vp = ctypes.c_void_p
        L.oracle_act_quant.restype = None
        L.oracle_act_quant.argtypes = [ctypes.c_int, vp, vp, vp, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_float, ctypes.c_float, vp, vp, vp, vp]
        L.oracle_gemm_w4a4_f16.restype = None
        L.oracle_gemm_w4a4_f16.argtypes = [vp] * 9 + [ctypes.c_long] * 3
        L.oracle_gemm_w4a4_f32.restype = None
        L.oracle_gemm_w4a4_f32.argtypes = [vp] * 9 + [ctypes.c_long] * 3
        L.oracle_gemm_w4a4_f16_split.restype = None
        L.oracle_gemm_w4a4_f16_split.argtypes = [vp] * 9 + [ctypes.c_long] * 3 + [ctypes.c_int]
        L.oracle_gemv_w4a4_f16_lanes.restype = None
        L.oracle_gemv_w4a4_f16_lanes.argtypes = [vp] * 9 + [ctypes.c_long] * 3
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def act_quant(op, x, b, idx, sim, clip, eps=0.0):
    x = np.ascontiguousarray(x, dtype=np.float16)
    M, H = x.shape
    b = None if b is None else np.ascontiguousarray(b, dtype=np.float16)
    idx = None if idx is None else np.ascontiguousarray(idx, dtype=np.int16)
    G = (H - 128) // 128
    q4 = np.empty((M, H - 128), np.int8); s4 = np.empty((M, G), np.float16)
    q8 = np.empty((M, 128), np.int8); s8 = np.empty((M,), np.float16)
    lib().oracle_act_quant(op, _p(x), _p(b), _p(idx), M, H, int(sim), float(clip), float(eps), _p(q4), _p(s4), _p(q8), _p(s8))
    return dict(q4=q4, s4=s4, q8=q8, s8=s8)


def gemm(A4, B4, sA_GM, sB, A8, B8, sA8, sB8, fp32=False, nsplit=1):
    """Packed operands, sA_GM plain [G, M].  Returns float16 [M, N] under the C-ABI arithmetic contract
    (fp32=True: the FP32 accumulators before the final rounding; nsplit=8: the decode-batch kernel's summation order;
    nsplit=-2 / -4: the two- / four-K-group tile kernels' -- K steps in 2 / 4 ranges, partial sums added in order;
    nsplit="lanes": the one-token dot-product kernel's -- groups dealt to 16 quad leaders, 64-lane butterfly, keeper last)."""
    A4 = np.ascontiguousarray(A4, np.uint8); B4 = np.ascontiguousarray(B4, np.uint8)
    M, N = A4.shape[0], B4.shape[0]
    K = A4.shape[1] * 2 + 128
    arrs = [A4, B4, np.ascontiguousarray(sA_GM, np.float16), np.ascontiguousarray(sB, np.float16),
            np.ascontiguousarray(A8, np.int8), np.ascontiguousarray(B8, np.int8),
            np.ascontiguousarray(sA8, np.float16), np.ascontiguousarray(sB8, np.float16)]
    if fp32:
        D = np.empty((M, N), np.float32)
        lib().oracle_gemm_w4a4_f32(*[_p(a) for a in arrs], _p(D), M, N, K)
        return D
    D = np.empty((M, N), np.float16)
    if nsplit == "lanes":
        lib().oracle_gemv_w4a4_f16_lanes(*[_p(a) for a in arrs], _p(D), M, N, K)
        return D
    if nsplit > 1 or nsplit < 0:
        lib().oracle_gemm_w4a4_f16_split(*[_p(a) for a in arrs], _p(D), M, N, K, int(nsplit))
    else:
        lib().oracle_gemm_w4a4_f16(*[_p(a) for a in arrs], _p(D), M, N, K)
    return D
