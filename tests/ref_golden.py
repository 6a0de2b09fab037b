"""ctypes view of oracle/_ref/libatom_ref.so: the reference's own CPU golden functions (run_cpu_reorder_fp16_i4,
run_cpu_activate_fp16_i4, run_cpu_rmsnorm_fp16_i4), compiled from /root/reference by oracle/Makefile.  Checker side only.
The library exists where the build container made it (it travels to the GPU box as a built file); available() says so."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "_ref", "libatom_ref.so")
_lib = None


def available() -> bool:
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(_SO)
        vp, i = ctypes.c_void_p, ctypes.c_int
        L.ref_cpu_reorder_fp16_i4.restype = None
        L.ref_cpu_reorder_fp16_i4.argtypes = [vp, i, i, i, vp, vp, vp, vp, vp]
        L.ref_cpu_activate_fp16_i4.restype = None
        L.ref_cpu_activate_fp16_i4.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp]
        L.ref_cpu_rmsnorm_fp16_i4.restype = None
        L.ref_cpu_rmsnorm_fp16_i4.argtypes = [vp, vp, ctypes.c_float, i, i, i, vp, vp, vp, vp, vp]
        L.ref_cpu_append_paged_kv_i4.restype = None
        L.ref_cpu_append_paged_kv_i4.argtypes = [vp] * 5 + [i] * 6 + [vp] * 5
        L.ref_cpu_single_decode_i4.restype = None
        L.ref_cpu_single_decode_i4.argtypes = [vp] * 5 + [i, i, i, ctypes.c_float, ctypes.c_float, vp]
        L.ref_cpu_o4_epilogue.restype = None
        L.ref_cpu_o4_epilogue.argtypes = [vp, i, i, vp, vp]
        L.ref_scale_index.restype = i
        L.ref_scale_index.argtypes = [i]
        _lib = L
    return _lib


def scale_size(rows: int) -> int:
    """SCALE_SIZE_A (Reorder.cu:50)."""
    return rows // 16 * 64 + 64 - (1 - (rows % 16) // 8) * (8 - (rows % 8)) * 8


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _run(kind, x, b=None, w=None, idx=None, eps=0.0):
    """Returns dict(q4 int8 [M, K4] (unpacked), s4 f16 [M, G], q8 int8 [M, 128], s8 f16 [M]) read back from the reference's
    buffers: packed PackInt4 codes (low nibble = even element) and the replicated scale layout (first replica)."""
    x = np.ascontiguousarray(x, dtype=np.float16)
    M, H = x.shape
    G = H // 128 - 1
    ld = scale_size(M)
    o8 = np.zeros((M, 128), np.int8)
    o4 = np.zeros((M, (H - 128) // 2), np.uint8)
    s8 = np.zeros(ld, np.float16)
    s4 = np.zeros(G * ld, np.float16)
    L = lib()
    if kind == "reorder":
        idx = np.ascontiguousarray(idx, dtype=np.int16)
        L.ref_cpu_reorder_fp16_i4(_p(x), 128, H, M, _p(idx), _p(o8), _p(o4), _p(s8), _p(s4))
    elif kind == "activate":
        b = np.ascontiguousarray(b, dtype=np.float16)
        L.ref_cpu_activate_fp16_i4(_p(x), _p(b), 128, H, M, _p(o8), _p(o4), _p(s8), _p(s4))
    else:
        idx = np.ascontiguousarray(idx, dtype=np.int16)
        w = np.ascontiguousarray(w, dtype=np.float16)
        L.ref_cpu_rmsnorm_fp16_i4(_p(x), _p(w), float(eps), 128, H, M, _p(idx), _p(o8), _p(o4), _p(s8), _p(s4))
    rows = np.array([L.ref_scale_index(r) for r in range(M)])
    lo = (o4 & 0xF).astype(np.int8)
    hi = (o4 >> 4).astype(np.int8)
    q4 = np.stack([np.where(lo >= 8, lo - 16, lo), np.where(hi >= 8, hi - 16, hi)], axis=-1).reshape(M, H - 128).astype(np.int8)
    s4p = s4.reshape(G, ld)[:, rows].T.copy()
    # all four replicas of a scale must agree (rows + 2k)
    for k in range(1, 4):
        assert np.array_equal(s4.reshape(G, ld)[:, rows + 2 * k].T, s4p)
    return dict(q4=q4, s4=s4p, q8=o8, s8=s8[rows].copy(), raw_s4=s4, raw_s8=s8, packed4=o4)


def reorder(x, idx):
    return _run("reorder", x, idx=idx)


def activate(a, b):
    return _run("activate", a, b=b)


def rmsnorm(x, w, eps, idx):
    return _run("rmsnorm", x, w=w, idx=idx, eps=eps)


def o4_epilogue(c32):
    """The u4-output GEMM's epilogue as the reference CODE computes it (DenseLayerGEMM_i4_o4.cu:722-786 applied around the
    reference's own local_max_min, :72-80; oracle/ref/wrap.cpp) on FP32 sums [M, N] -> (u8 [M, N/2], f16 (scale, zero) [M, N/128, 2]).
    The reference casts round(...) to int8_t with no clamp (:766-767): out of the int8 range that is undefined behaviour on the host
    (and a saturating cvt on the device), so callers compare only where in_range() holds."""
    c32 = np.ascontiguousarray(c32, dtype=np.float32)
    M, N = c32.shape
    D = np.zeros((M, N // 2), np.uint8)
    sz = np.zeros((M, N // 128, 2), np.float16)
    lib().ref_cpu_o4_epilogue(_p(c32), M, N, _p(D), _p(sz))
    return D, sz


def o4_in_range(c32):
    """[M, N/128] bool: groups whose reference-code quotients (x + zero) * r all lie inside the int8 range (see o4_epilogue)."""
    g = np.asarray(c32, np.float32).reshape(c32.shape[0], -1, 128)
    a = np.abs(g)
    mx, mn = a.max(-1), a.min(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.float32(1) / ((mx - mn) / np.float32(15)).astype(np.float32)
        q = ((g - mn[..., None]).astype(np.float32) * r[..., None]).astype(np.float32)
    return np.isfinite(q).all(-1) & (np.abs(q) < 127).all(-1)


# ---- INT4 paged KV cache: the reference's CPU restatements (kernels/src/flashinfer/cpu_reference.h)
def append_paged_kv_i4(data, param, indptr, indices, last_page_offset, k, v, k_param, v_param, layer, append_indptr):
    """In place on data u8 [pages, L, 2, N, P, D/2] / param f16 [pages, L, 2, N, P, 2]; k, v u8 [T, N, D/2]; k_param, v_param f16 [T, N, 2]."""
    assert data.flags.c_contiguous and param.flags.c_contiguous and data.dtype == np.uint8 and param.dtype == np.float16
    pages, Lr, _, N, P, Dh = data.shape
    c32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    ip, ix, lp, ap = c32(indptr), c32(indices), c32(last_page_offset), c32(append_indptr)
    k, v = np.ascontiguousarray(k, np.uint8), np.ascontiguousarray(v, np.uint8)
    kp, vpm = np.ascontiguousarray(k_param, np.float16), np.ascontiguousarray(v_param, np.float16)
    lib().ref_cpu_append_paged_kv_i4(_p(data), _p(param), _p(ip), _p(ix), _p(lp), Lr, int(layer), N, P, Dh * 2, len(lp), _p(k), _p(v),
                                     _p(kp), _p(vpm), _p(ap))


def single_decode_i4(q16, k, v, k_param, v_param, theta=1e4):
    """q f16 [N, D]; k, v u8 [S, N, D/2]; k_param, v_param f16 [S, N, 2] -> float32 [N, D] (llama RoPE on q at S - 1 and on every key)."""
    q16 = np.ascontiguousarray(q16, np.float16)
    k, v = np.ascontiguousarray(k, np.uint8), np.ascontiguousarray(v, np.uint8)
    kp, vpm = np.ascontiguousarray(k_param, np.float16), np.ascontiguousarray(v_param, np.float16)
    S, N, Dh = k.shape
    out = np.empty((N, Dh * 2), np.float32)
    lib().ref_cpu_single_decode_i4(_p(q16), _p(k), _p(v), _p(kp), _p(vpm), S, N, Dh * 2, 1.0, float(theta), _p(out))
    return out
