"""Native op wrappers with the call surface of the reference's ``punica.ops``
(/root/reference/e2e/punica-atom/punica/ops/__init__.py:137-219): same names, same positional arguments,
same return tuples and output shapes/dtypes.  Each call allocates its outputs with ``torch.empty`` and
enqueues one HIP kernel of libatom_hip.so on the current stream.

Keyword-only extras (defaults reproduce the reference CUDA kernels exactly):
  quant_mode   "kernel" (Reorder.cuh:137-178 arithmetic) | "sim" (model/quant.py arithmetic)
  clip         clip ratio applied to the INT4 groups' absmax (1.0 = none, the kernels' behaviour)
  scale_layout "ref" (ldmatrix-replicated layout, scale_size(M) halves per group) | "plain" ([G, M])
  return_dequant  also return the de-quantised FP16 tensor (what quantize_activation_wrapper returns)
  wide_codes   (activation ops) / a_wide (GEMM): the native activation format of include/atom_hip.h -- o_norms is
               int8 [bs, H-128] = code*16 with the even/odd channels of every 32-channel block de-interleaved, i.e.
               already the INT8 MFMA operand; the prefill GEMM then skips 2/3 of its widening instructions.
               wide_codes="f6" / a_wide="f6": the BF6 group-major format [G][f6_rows(bs)][104] of the block-scaled-MFMA
               prefill kernel; the GEMM then also wants the weight from repack_weight_f6
"""
from __future__ import annotations

import collections
import weakref

import torch

from . import _lib as L

GROUP_SIZE = 128

_MODES = {"kernel": L.QUANT_KERNEL, "sim": L.QUANT_SIM}
_LAYOUTS = {"ref": L.SCALE_LAYOUT_REF, "plain": L.SCALE_LAYOUT_PLAIN}


def scale_size(x: int) -> int:
    """punica/ops/__init__.py:137-138 (SCALE_SIZE_A, Reorder.cuh:50)."""
    return ((x) // 16 * 64 + 64 - (1 - (x % 16) // 8) * (8 - (x % 8)) * 8)


def _ld(rows: int, layout: str) -> int:
    return scale_size(rows) if layout == "ref" else rows


def _require_cuda_half(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise L.AtomHipError(f"{name} must live on the GPU: the Atom W4A4 path has no CPU fallback")
    if t.dtype != torch.float16:
        raise TypeError(f"{name} must be float16, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


def f6_rows(rows: int) -> int:
    """Rows per group of an F6 operand buffer (include/atom_hip.h, ATOM_AB_F6): rows rounded up to 256."""
    return (rows + 255) // 256 * 256


def _alloc_act_outputs(bs, hidden_dim, device, layout, return_dequant, wide=False):
    # empty, like the reference's wrappers (punica/ops/__init__.py:189-196): the replicated layout has slots no row ever
    # writes, and no kernel here reads them (rows are clamped to M - 1); zero-filling them cost two fill launches per op
    alloc = torch.empty
    o_outlier = torch.empty((bs, GROUP_SIZE), dtype=torch.int8, device=device)
    if wide == "f6":        # [G][rows_pad][104] BF6 streams + in-row scales; pad rows are never read into a result
        o_norms = torch.empty((hidden_dim // GROUP_SIZE - 1, f6_rows(bs), L.F6_PITCH), dtype=torch.uint8, device=device)
    else:
        o_norms = torch.empty((bs, (hidden_dim - GROUP_SIZE) // (1 if wide else 2)), dtype=torch.int8, device=device)
    outlier_scales = alloc((_ld(bs, layout),), dtype=torch.float16, device=device)
    norm_scales = alloc((hidden_dim // GROUP_SIZE - 1, _ld(bs, layout)), dtype=torch.float16, device=device)
    xq = torch.empty((bs, hidden_dim), dtype=torch.float16, device=device) if return_dequant else None
    return o_outlier, o_norms, outlier_scales, norm_scales, xq


def _ret(o_outlier, o_norms, outlier_scales, norm_scales, xq):
    if xq is None:
        return o_outlier, o_norms, outlier_scales, norm_scales
    return o_outlier, o_norms, outlier_scales, norm_scales, xq


def _mode(quant_mode, wide_codes):
    """wide_codes: False (packed nibbles, the reference format) | True (int8 code*16) | "f6" (BF6 group-major)."""
    flag = L.QUANT_F6_CODES if wide_codes == "f6" else (L.QUANT_WIDE_CODES if wide_codes else 0)
    return _MODES[quant_mode] | flag


def activate_fp16_i4(a: torch.Tensor, b: torch.Tensor, *, quant_mode="kernel", clip=1.0, scale_layout="ref",
                     return_dequant=False, wide_codes=False):
    """quant(silu(a) * b) -> (o_outlier i8[bs,128], o_norms i8[bs,(H-128)/2], outlier_scales, norm_scales).
    Reference: punica/ops/__init__.py:141-156 -> run_activate_fp16_i4 (Activate.cuh:194-217).  Any H % 128 == 0
    (the reference is hard-instantiated for 11008, punica_ops.cc:76)."""
    _require_cuda_half(a, "a")
    _require_cuda_half(b, "b")
    bs, hidden_dim = a.shape
    assert b.shape == a.shape
    outs = _alloc_act_outputs(bs, hidden_dim, a.device, scale_layout, return_dequant, wide_codes)
    st = L.lib().atom_silu_mul_quant_f16(a.data_ptr(), b.data_ptr(), bs, hidden_dim, _mode(quant_mode, wide_codes), clip,
                                          _LAYOUTS[scale_layout], outs[0].data_ptr(), outs[1].data_ptr(),
                                          outs[2].data_ptr(), outs[3].data_ptr(), L.ptr(outs[4]),
                                          L.current_stream(a.device))
    L.check(st, "atom_silu_mul_quant_f16")
    return _ret(*outs)


def rmsnorm_fp16_i4(hidden_states: torch.Tensor, weight: torch.Tensor, reorder_index: torch.Tensor, eps: float, *,
                    quant_mode="kernel", clip=1.0, scale_layout="ref", return_dequant=False, wide_codes=False):
    """quant(index_select(RMSNorm(x)*w, reorder_index)).  Reference: punica/ops/__init__.py:183-200 ->
    run_rmsnorm_fp16_i4 (RMSNorm.cuh:255-285).  reorder_index is int16 [H] as in the reference."""
    _require_cuda_half(hidden_states, "hidden_states")
    _require_cuda_half(weight, "weight")
    assert reorder_index.dtype == torch.int16 and reorder_index.is_cuda
    bs, hidden_dim = hidden_states.shape
    outs = _alloc_act_outputs(bs, hidden_dim, hidden_states.device, scale_layout, return_dequant, wide_codes)
    st = L.lib().atom_rmsnorm_reorder_quant_f16(hidden_states.data_ptr(), weight.data_ptr(), float(eps),
                                                 reorder_index.data_ptr(), bs, hidden_dim,
                                                 _mode(quant_mode, wide_codes), clip,
                                                 _LAYOUTS[scale_layout], outs[0].data_ptr(), outs[1].data_ptr(),
                                                 outs[2].data_ptr(), outs[3].data_ptr(), L.ptr(outs[4]),
                                                 L.current_stream(hidden_states.device))
    L.check(st, "atom_rmsnorm_reorder_quant_f16")
    return _ret(*outs)


def add_rmsnorm_fp16_i4(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, reorder_index: torch.Tensor,
                        eps: float, *, inplace=False, quant_mode="kernel", clip=1.0, scale_layout="ref",
                        return_dequant=False, wide_codes=False):
    """NEW (no reference op): s = x + residual (fp16), then rmsnorm_fp16_i4(s, ...) -- the residual add
    of the decoder layer (llama.py:266-282) fused into the following RMSNorm-quant kernel.  Returns (s, *quant outputs);
    ``inplace`` writes s over ``residual``."""
    _require_cuda_half(x, "x")
    _require_cuda_half(residual, "residual")
    _require_cuda_half(weight, "weight")
    assert x.shape == residual.shape and reorder_index.dtype == torch.int16 and reorder_index.is_cuda
    bs, hidden_dim = x.shape
    s = residual if inplace else torch.empty_like(residual)
    outs = _alloc_act_outputs(bs, hidden_dim, x.device, scale_layout, return_dequant, wide_codes)
    st = L.lib().atom_add_rmsnorm_reorder_quant_f16(x.data_ptr(), residual.data_ptr(), s.data_ptr(), weight.data_ptr(),
                                                     float(eps), reorder_index.data_ptr(), bs, hidden_dim,
                                                     _mode(quant_mode, wide_codes), clip, _LAYOUTS[scale_layout],
                                                     outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                                                     outs[3].data_ptr(), L.ptr(outs[4]), L.current_stream(x.device))
    L.check(st, "atom_add_rmsnorm_reorder_quant_f16")
    return (s,) + tuple(_ret(*outs))


def reorder_fp16_i4(hidden_states: torch.Tensor, reorder_index, *, quant_mode="kernel", clip=1.0,
                    scale_layout="ref", return_dequant=False, wide_codes=False):
    """quant(index_select(x, reorder_index)).  Reference: punica/ops/__init__.py:203-219 ->
    run_reorder_fp16_i4 (Reorder.cuh:205-228).  ``reorder_index=None`` quantises x in its given channel order
    (the tail of quantize_activation_wrapper, model/quant.py:187-231)."""
    _require_cuda_half(hidden_states, "hidden_states")
    if reorder_index is not None:
        assert reorder_index.dtype == torch.int16 and reorder_index.is_cuda
    bs, hidden_dim = hidden_states.shape
    outs = _alloc_act_outputs(bs, hidden_dim, hidden_states.device, scale_layout, return_dequant, wide_codes)
    st = L.lib().atom_reorder_quant_f16(hidden_states.data_ptr(), L.ptr(reorder_index), bs, hidden_dim,
                                         _mode(quant_mode, wide_codes), clip, _LAYOUTS[scale_layout], outs[0].data_ptr(),
                                         outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), L.ptr(outs[4]),
                                         L.current_stream(hidden_states.device))
    L.check(st, "atom_reorder_quant_f16")
    return _ret(*outs)


_WS = {}
_WS_RETIRED = []               # buffers replaced during a graph capture: launches captured earlier still point at them


def _workspace(device, nbytes):
    """Scratch for split-K partial sums / re-coded operands / decode partials, per (device, stream): calls on one stream reuse it
    in stream order; calls on different streams never share a buffer.  Grown on demand (the old buffer is returned to the caching
    allocator, which keeps it alive for work already queued on its stream).  During HIP-graph capture a too small buffer is not
    freed -- launches captured earlier (also of an earlier graph captured on the same stream) reference it -- but retired, and
    the new one comes out of the capturing graph's memory pool like every other tensor allocated during capture.  Nothing is ever
    expected to SURVIVE in this buffer from one call to the next (round 3 kept a weight's re-coded form at its head and trusted
    that no other call had touched it -- unsafe under graph capture, and useless for a model whose projections share the buffer)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        if t is not None and torch.cuda.is_current_stream_capturing():
            _WS_RETIRED.append(t)
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _WS[key] = t
    return t


# ---- re-coded (F6) forms of PACKED weights that meet prefill-size batches through the reference-format entry points -------------------
# A caller of the reference's operator surface hands dense_layer_gemm_i4_fp16 the packed INT4 weight on every call (llama.py:61-68).
# From 257 rows on the fastest kernels want the F6 form; re-coding the weight on every call costs as much as re-coding the activation
# (12 vs 6 us at 4096 x 4096).  So the F6 form of every weight seen here is kept -- per WEIGHT (its own tensor, nothing shares it), keyed
# by the storage addresses and version counters of the packed weight and its scales, least recently used first out under a byte cap.
# A Llama-7B's 224 projections are 5.5 GB in this form (0.84 B per weight) next to 288 GB of HBM.
#   * an entry dies with the packed weight: weakref finalizers on the storages of the weight and of its scales drop it (so a cached
#     address is never handed to another tensor while its entry lives, and deleting a model frees its BF6 forms -- rounds 3-4 held
#     strong references);
#   * the byte cap defaults to an eighth of the device's memory, at most 32 GiB (set_weight_f6_cache_bytes changes it);
#   * an entry is complete before its first use on ANY stream: the stream that re-coded the weight is synchronised once, at creation;
#   * tensors made under torch.inference_mode() have no version counter: they count as version -1 (they cannot be written in place);
#   * a weight rewritten in place bumps its version counter -> miss (writes through a `.data` / `.detach()` alias do NOT bump it:
#     call forget_weight_f6(b) after such a write);
#   * HIP-graph capture: nothing executes during capture, so an entry is never CREATED there (the call takes the workspace route and
#     re-codes both operands inside the captured graph -- always right); a HIT during capture pins the entry for good, because the
#     captured launch keeps pointing at it.
_F6W = collections.OrderedDict()      # key -> [f6s tensor, finalizers, bytes, pinned]
_F6W_STATE = {"limit": None, "bytes": 0, "hits": 0, "misses": 0}


def _f6w_limit(device):
    if _F6W_STATE["limit"] is None:
        _F6W_STATE["limit"] = min(32 << 30, torch.cuda.get_device_properties(device).total_memory // 8)
    return _F6W_STATE["limit"]


def _version_of(t):
    try:
        return t._version
    except RuntimeError:                                     # inference tensors do not track a version counter
        return -1


def _drop_f6w(key):
    e = _F6W.pop(key, None)
    if e is not None:
        _F6W_STATE["bytes"] -= e[2]
        for f in e[1]:                                       # (a no-op for the finalizer that is calling us)
            f.detach()


def set_weight_f6_cache_bytes(limit: int):
    """Byte cap of the re-coded-weight cache (0 disables it: every call re-codes both operands in the workspace)."""
    _F6W_STATE["limit"] = int(limit)
    _evict_f6w()


def clear_weight_f6_cache():
    for e in _F6W.values():
        for f in e[1]:
            f.detach()
    _F6W.clear()
    _F6W_STATE.update(bytes=0, hits=0, misses=0)


def forget_weight_f6(b: torch.Tensor, b_scale: torch.Tensor = None):
    """Drop the cached F6 form(s) of the packed weight ``b`` (after writing it through an alias that keeps the version counter --
    ``.data`` / ``.detach()`` writes, inference tensors).  Pass the weight's scale tensor as well after such a write to IT: the
    channel-pair tag of ``_pairs_flag`` lives on that tensor object and is keyed by the same version counter."""
    for key in [k for k in _F6W if k[1] == b.data_ptr() and k[0] == b.device]:
        _drop_f6w(key)
    for t in (b, b_scale):
        if t is not None and getattr(t, "_atom_pairs", None) is not None:
            try:
                del t._atom_pairs
            except AttributeError:
                pass


def _evict_f6w():
    for key in list(_F6W):
        if _F6W_STATE["limit"] is None or _F6W_STATE["bytes"] <= _F6W_STATE["limit"]:
            break
        if not _F6W[key][3]:
            for f in _F6W[key][1]:
                f.detach()
            _drop_f6w(key)


def _weight_f6s(b, b_scale, n, k):
    """The cached F6 form (codes + float32 scales, ATOM_B_F6S) of a packed weight, or None when there is none and none may be made."""
    capturing = torch.cuda.is_current_stream_capturing()
    key = (b.device, b.data_ptr(), _version_of(b), b_scale.data_ptr(), _version_of(b_scale), n, k)
    e = _F6W.get(key)
    if e is not None:
        _F6W.move_to_end(key)
        _F6W_STATE["hits"] += 1
        if capturing:
            e[3] = True
        return e[0]
    if capturing or _f6w_limit(b.device) <= 0:
        return None
    _F6W_STATE["misses"] += 1
    g = k // GROUP_SIZE - 1
    f6 = repack_weight_f6(b.view(torch.uint8), b_scale.reshape(-1)[:g * n])      # b_scale is read flat as [G][N] whatever its shape
    nbytes = f6.atom_f6s.numel()
    if nbytes > _F6W_STATE["limit"]:
        return f6
    torch.cuda.current_stream(b.device).synchronize()        # complete before any other stream (or a graph captured later) reads it
    fin = tuple(weakref.finalize(st, _drop_f6w, key) for st in (b.untyped_storage(), b_scale.untyped_storage()))
    if not all(f.alive for f in fin):
        # a torch whose untyped_storage() hands out a fresh wrapper per call: the finalizers have fired already and the entry could
        # never be dropped -- a recycled address would then hit a stale weight.  Do not cache (the call re-codes in the workspace).
        for f in fin:
            f.detach()
        return f6
    _F6W[key] = [f6, fin, nbytes, False]
    _F6W_STATE["bytes"] += nbytes
    _evict_f6w()
    return f6


def gemm_recodes_cached(m, n, k):
    """True where a packed activation of m rows should be re-coded to BF6 for a weight [n, k] whose BF6 form is at hand (the rule of
    ATOM_WS_WEIGHT_CACHED, atom_gemm_w4a4_ws_recodes_cached in include/atom_hip.h): what modules that own their weight's BF6 form ask,
    so that it is not made a second time in this module's per-weight cache."""
    return bool(L.lib().atom_gemm_w4a4_ws_recodes_cached(int(m), int(n), int(k)))


def _gemm_dims(a, b, a_keeper, a_wide=False, b_keeper=None):
    if a_wide == "f6":                                             # [G][rows_pad][104]: sizes come from the keepers
        return a_keeper.size(0), b_keeper.size(0), a.size(0) * GROUP_SIZE + a_keeper.size(1)
    m = a.size(0)
    n = b.size(0)
    k = a.size(1) * (1 if a_wide else 2) + a_keeper.size(1)       # punica_ops.cc:228-241
    return m, n, k


def dense_layer_gemm_i4_fp16(a, b, a_scale, b_scale, a_keeper, b_keeper, a_keeper_scale, b_keeper_scale, *,
                             scale_layout="ref", a_wide=False):
    """d[M,N] fp16 = W4A4 group-128 GEMM + INT8 keeper.  Reference: punica/ops/__init__.py:159-167 ->
    DenseLayerGEMM_i4<nv_half> (DenseLayerGEMM_i4.cu:722-791).  b_scale is read flat as [G][N] and b_keeper_scale
    as [N], exactly like the reference kernel (Dense_layer_gemm_i4_o16.cuh:497), whatever the tensor's shape."""
    for t in (a, b, a_scale, b_scale, a_keeper, b_keeper, a_keeper_scale, b_keeper_scale):
        if not t.is_cuda:
            raise L.AtomHipError("all GEMM operands must live on the GPU: no CPU fallback")
    m, n, k = _gemm_dims(a, b, a_keeper, a_wide, b_keeper)
    if a_wide == "f6":
        assert a.shape == (k // GROUP_SIZE - 1, f6_rows(m), L.F6_PITCH) and b.shape == (a.shape[0], f6_rows(n), L.F6_PITCH)
    d = torch.empty((m, n), dtype=torch.float16, device=a.device)
    lib = L.lib()
    # > 0 for skinny shapes that gain from split-K, and for packed operands of prefill size (re-coded to F6 in the workspace);
    # operands that are F6 already need none (the F6 kernels take no workspace)
    ws_bytes = lib.atom_gemm_w4a4_workspace_bytes(m, n, k) if not (a_wide == "f6" or (a_wide and m >= 2048)) else 0
    # packed operands of prefill size (the re-coding route): with the weight's F6 form cached (per weight, _weight_f6s) only the
    # activation is re-coded, into a fresh tensor, and the F6 kernel runs on the two -- the kernel and the bits of the workspace route
    # (with the weight's BF6 form cached the route starts at 129 rows, and at 17 where the decode-batch kernel does not take the shape
    # -- only the activation is re-coded and the mid-size-batch BF6 kernel beats what runs otherwise: ATOM_WS_WEIGHT_CACHED's rule,
    # atom_gemm_w4a4_ws_recodes_cached in include/atom_hip.h)
    if ws_bytes and not a_wide and lib.atom_gemm_w4a4_ws_recodes_cached(m, n, k):
        f6w = _weight_f6s(b, b_scale, n, k)
        if f6w is not None:
            a6 = repack_act_f6(a.view(torch.uint8), a_scale, scale_layout=scale_layout)
            return dense_layer_gemm_i4_fp16(a6, f6w, a_scale, b_scale, a_keeper, b_keeper, a_keeper_scale, b_keeper_scale,
                                            scale_layout=scale_layout, a_wide="f6")
    ws = _workspace(a.device, ws_bytes) if ws_bytes else None
    st = lib.atom_gemm_w4a4_f16_ws(a.data_ptr(), b.data_ptr(), a_scale.data_ptr(), b_scale.data_ptr(),
                                   a_keeper.data_ptr(), b_keeper.data_ptr(), a_keeper_scale.data_ptr(),
                                   b_keeper_scale.data_ptr(), d.data_ptr(), m, n, k, GROUP_SIZE, GROUP_SIZE,
                                   _LAYOUTS[scale_layout] | (L.AB_F6 if a_wide == "f6" else (L.A_WIDE if a_wide else 0))
                                   | (L.B_F6S if a_wide == "f6" and getattr(b, "atom_f6s", None) is not None else 0)
                                   | (L.B_SCALE_PAIRS if (getattr(b, "atom_pairs", False) if a_wide == "f6" else _pairs_flag(b_scale, n)) else 0),
                                   L.ptr(ws), ws_bytes,
                                   L.current_stream(a.device))
    L.check(st, "atom_gemm_w4a4_f16_ws")
    return d


def dense_layer_gemm_i4_o4(a, b, a_scale, b_scale, a_keeper, b_keeper, a_keeper_scale, b_keeper_scale, *,
                           scale_layout="ref", use_workspace=True, ref_extrema=False):
    """Same GEMM, output asymmetric-quantised to u4 per 128-column group: (d u8[M,N/2], d_scale f16[M,N/128*2]).
    Reference: punica/ops/__init__.py:170-180 -> DenseLayerGEMM_i4_o4 (DenseLayerGEMM_i4_o4.cu:808-856).
    Decode batches (the shapes of the decode-batch GEMM) go through atom_gemm_w4a4_o4_ws with the cached workspace (weight-streaming kernel + u4
    epilogue launch); ``use_workspace=False`` forces the tile kernel.  ``ref_extrema=True``: the epilogue exactly as the reference
    CODE computes it (extrema of |x|, 4-bit wrap instead of a clamp: ATOM_O4_REF_EXTREMA in include/atom_hip.h)."""
    m, n, k = _gemm_dims(a, b, a_keeper)
    assert n % 128 == 0
    d = torch.empty((m, n // 2), dtype=torch.uint8, device=a.device)
    d_scale = torch.empty((m, n // 128 * 2), dtype=torch.float16, device=a.device)
    lib = L.lib()
    ws_bytes = lib.atom_gemm_w4a4_o4_workspace_bytes(m, n, k) if use_workspace else 0
    ws = _workspace(a.device, ws_bytes) if ws_bytes else None
    st = lib.atom_gemm_w4a4_o4_ws(a.data_ptr(), b.data_ptr(), a_scale.data_ptr(), b_scale.data_ptr(),
                                  a_keeper.data_ptr(), b_keeper.data_ptr(), a_keeper_scale.data_ptr(),
                                  b_keeper_scale.data_ptr(), d.data_ptr(), d_scale.data_ptr(), m, n, k, GROUP_SIZE,
                                  GROUP_SIZE, _LAYOUTS[scale_layout] | (L.O4_REF_EXTREMA if ref_extrema else 0), L.ptr(ws), ws_bytes,
                                  L.current_stream(a.device))
    L.check(st, "atom_gemm_w4a4_o4_ws")
    return d, d_scale


def decode_gemm_fits(m: int, n: int, k: int) -> bool:
    """True when (m, n, k) is a shape the decode-batch GEMM takes, i.e. dense_layer_gemm_i4_f32 is available."""
    return m >= 1 and n % 128 == 0 and L.lib().atom_gemm_w4a4_o4_workspace_bytes(m, n, k) > 0


def dense_layer_gemm_i4_f32(a, b, a_scale, b_scale, a_keeper, b_keeper, a_keeper_scale, b_keeper_scale, *,
                            scale_layout="ref"):
    """NEW (no reference counterpart): the FP32 sums of the GEMM, float [M, N], for decode batches (decode_gemm_fits) --
    what the u4 epilogue quantises; feeds quant_append_kv_i4."""
    m, n, k = _gemm_dims(a, b, a_keeper)
    d = torch.empty((m, n), dtype=torch.float32, device=a.device)
    st = L.lib().atom_gemm_w4a4_f32(a.data_ptr(), b.data_ptr(), a_scale.data_ptr(), b_scale.data_ptr(),
                                    a_keeper.data_ptr(), b_keeper.data_ptr(), a_keeper_scale.data_ptr(),
                                    b_keeper_scale.data_ptr(), d.data_ptr(), m, n, k, GROUP_SIZE, GROUP_SIZE,
                                    _LAYOUTS[scale_layout], L.current_stream(a.device))
    L.check(st, "atom_gemm_w4a4_f32")
    return d


def fuse_projection_weights(mods):
    """Offline, once per layer: the packed weights of 2-3 ``LinearInt4`` modules that read the same activation (q / k / v; gate /
    up) as ONE operand for dense_layer_gemm_i4_multi, concatenated along the output features.  The codes are not duplicated: the
    modules' ``weight_int4`` / ``weight_int8`` parameters become views of the fused buffers (a later in-place load writes through;
    a re-assigned parameter changes the key and the operand is rebuilt; serialisers that refuse tensors sharing storage, e.g.
    safetensors' save_file, want ``.clone()``d parameters).  The scales (31 x N halves per projection) are copied.
    Returns a dict; ``fused_key(mods)`` tells whether it is current."""
    packs = [m.packed() for m in mods]
    n = packs[0][0].shape[0]
    assert all(pk[0].shape == packs[0][0].shape for pk in packs) and n % 16 == 0
    b4 = torch.cat([pk[0].view(torch.uint8) for pk in packs], 0).contiguous()
    b8 = torch.cat([pk[1] for pk in packs], 0).contiguous()
    g = packs[0][2].shape[0]
    sb = torch.cat([pk[2].reshape(g, n) for pk in packs], 1).contiguous()
    sb8 = torch.cat([pk[3].reshape(-1) for pk in packs], 0).contiguous()
    for i, m in enumerate(mods):                                   # the parameters now alias the fused storage
        m.weight_int4.data = b4[i * n:(i + 1) * n]
        m.weight_int8.data = b8[i * n:(i + 1) * n]
    return {"b4": b4, "b8": b8, "sb": sb, "sb8": sb8, "n_seg": n, "nseg": len(mods), "k": b4.shape[1] * 2 + GROUP_SIZE,
            "key": fused_key(mods)}


def fused_key(mods):
    return tuple((m.weight_int4.data_ptr(), m.weight_int4._version, m.weight_int8.data_ptr(), m.weight_int8._version,
                  m.scale_int4.data_ptr(), m.scale_int4._version, m.scale_int8.data_ptr(), m.scale_int8._version) for m in mods)


def multi_gemm_fits(m: int, n_seg: int, nseg: int, k: int) -> bool:
    return bool(L.lib().atom_gemm_w4a4_multi_fits(int(m), int(n_seg), int(nseg), int(k)))


def dense_layer_gemm_i4_multi(a, a_scale, a_keeper, a_keeper_scale, fused, *, f32_mask=0, add=None, scale_layout="ref"):
    """NEW (decode batches; no reference counterpart): the projections in ``fused`` (fuse_projection_weights) applied to one
    activation operand in ONE launch.  Returns a tuple of [M, n_seg] tensors, fp16 or (bits of ``f32_mask``) the FP32 sums;
    ``add`` (fp16 [M, n_seg]) is added to the first output the way torch adds two half tensors (the residual add).  Bit-identical to
    dense_layer_gemm_i4_f32 / the decode-batch kernel per projection (+ torch's add)."""
    m = a.size(0)
    n, nseg, k = fused["n_seg"], fused["nseg"], fused["k"]
    assert a.size(1) * 2 + a_keeper.size(1) == k
    outs = [torch.empty((m, n), dtype=torch.float32 if (f32_mask >> i) & 1 else torch.float16, device=a.device) for i in range(nseg)]
    if add is not None:
        _require_cuda_half(add, "add")
        assert add.shape == (m, n) and add.is_contiguous()
    st = L.lib().atom_gemm_w4a4_multi(a.data_ptr(), fused["b4"].data_ptr(), a_scale.data_ptr(), fused["sb"].data_ptr(), a_keeper.data_ptr(),
                                      fused["b8"].data_ptr(), a_keeper_scale.data_ptr(), fused["sb8"].data_ptr(), outs[0].data_ptr(),
                                      outs[1].data_ptr() if nseg > 1 else None, outs[2].data_ptr() if nseg > 2 else None,
                                      int(f32_mask), L.ptr(add), m, n, nseg, k, GROUP_SIZE, GROUP_SIZE, _LAYOUTS[scale_layout],
                                      L.current_stream(a.device))
    L.check(st, "atom_gemm_w4a4_multi")
    return tuple(outs)


_Q_OPS = {"reorder": L.Q_REORDER, "rmsnorm": L.Q_RMSNORM, "add_rmsnorm": L.Q_ADD_RMSNORM, "silu_mul": L.Q_SILU_MUL}


def multi_q_gemm_fits(q_op: str, m: int, n_seg: int, nseg: int, k: int) -> bool:
    """True when dense_layer_gemm_i4_multi_q(q_op, ...) takes the shape (the launcher's own predicate)."""
    return bool(L.lib().atom_gemm_w4a4_multi_q_fits(_Q_OPS[q_op], int(m), int(n_seg), int(nseg), int(k)))


def dense_layer_gemm_i4_multi_q(q_op: str, x, fused, *, x2=None, residual=None, reorder_index=None, eps=0.0, clip=1.0, f32_mask=0, add=None):
    """NEW (decode steps of one or two tokens; no reference counterpart): dense_layer_gemm_i4_multi with the quantiser that precedes
    it in the reference's call order INSIDE the launch -- ``q_op`` "reorder" (reorder_fp16_i4), "rmsnorm" (rmsnorm_fp16_i4, ``x2`` = the
    norm weight), "add_rmsnorm" (add_rmsnorm_fp16_i4: returns x + residual as well) or "silu_mul" (activate_fp16_i4, ``x2`` = the second
    factor); kernel-flavoured quantiser arithmetic.  ``x`` fp16 [M, K].  Returns (outs, residual_out).  The quantised operand is
    bit-identical to the quantiser op's, and the GEMM behind it sums in the order the SEPARATE entry points use for the token count
    (atom_gemm_w4a4_packed_order): the dot-product kernel's (64) for one token and for two with K > 4096 -- round 6, csrc/gemvq_w4a4.hip:
    one quantiser per CU in front of that kernel -- and the decode-batch kernel's (8) for two tokens with K <= 4096.  The outputs are
    therefore bit-identical to quantiser op + dense_layer_gemm_i4_multi throughout (tests/test_gpu_e2e.py); rounds 3-5 always ran the
    decode-batch kernel and were one fp16 ulp off where the separate path took the dot-product kernel."""
    _require_cuda_half(x, "x")
    code = _Q_OPS[q_op]
    m = x.size(0)
    n, nseg, k = fused["n_seg"], fused["nseg"], fused["k"]
    assert x.shape == (m, k) and x.is_contiguous()
    outs = [torch.empty((m, n), dtype=torch.float32 if (f32_mask >> i) & 1 else torch.float16, device=x.device) for i in range(nseg)]
    res_out = None
    if code == L.Q_ADD_RMSNORM:
        _require_cuda_half(residual, "residual")
        assert residual.shape == x.shape and residual.is_contiguous()
        res_out = torch.empty_like(x)                    # (never in place: every workgroup reads the residual, one writes the sum)
    if x2 is not None:
        _require_cuda_half(x2, "x2")
        assert x2.is_contiguous() and (x2.shape == x.shape if code == L.Q_SILU_MUL else x2.numel() == k)
    if reorder_index is not None:
        assert reorder_index.dtype == torch.int16 and reorder_index.numel() == k and reorder_index.is_cuda
    if add is not None:
        _require_cuda_half(add, "add")
        assert add.shape == (m, n) and add.is_contiguous()
    st = L.lib().atom_gemm_w4a4_multi_q(code, x.data_ptr(), L.ptr(x2), L.ptr(residual), L.ptr(res_out), L.ptr(reorder_index), float(eps),
                                        float(clip), fused["b4"].data_ptr(), fused["sb"].data_ptr(), fused["b8"].data_ptr(),
                                        fused["sb8"].data_ptr(), outs[0].data_ptr(), outs[1].data_ptr() if nseg > 1 else None,
                                        outs[2].data_ptr() if nseg > 2 else None, int(f32_mask), L.ptr(add), m, n, nseg, k,
                                        GROUP_SIZE, GROUP_SIZE, L.current_stream(x.device))
    L.check(st, "atom_gemm_w4a4_multi_q")
    return tuple(outs), res_out


def merge_q_gemm_fits(m: int, n_seg: int, nseg: int, k: int, splits: int) -> bool:
    """True when dense_layer_gemm_i4_merge_q takes the shape and the KV-split count (atom_gemm_w4a4_multi_merge_q_fits)."""
    return bool(L.lib().atom_gemm_w4a4_multi_merge_q_fits(int(m), int(n_seg), int(nseg), int(k), int(splits)))


def dense_layer_gemm_i4_merge_q(partials: torch.Tensor, splits: int, fused, *, reorder_index=None, clip=1.0, f32_mask=0, add=None):
    """NEW (round 6; decode steps of one or two tokens, no reference counterpart): dense_layer_gemm_i4_multi_q("reorder", ...) on the
    decode attention's output while that is still in KV-split form -- ``partials`` float32 [M, heads, splits, 130] as
    ``batch_decode_i4(..., merge=False)`` returns it --, the split merge running in front of the reorder quantiser inside the
    projection's launch.  Bit-identical to batch_decode_i4 (merged) -> dense_layer_gemm_i4_multi_q("reorder", ...).  Returns outs."""
    if not partials.is_cuda:
        raise L.AtomHipError("merge_q needs GPU tensors: no CPU fallback")
    n, nseg, k = fused["n_seg"], fused["nseg"], fused["k"]
    m = partials.size(0)
    assert partials.dtype == torch.float32 and partials.is_contiguous() and partials.shape == (m, k // 128, splits, 130)
    outs = [torch.empty((m, n), dtype=torch.float32 if (f32_mask >> i) & 1 else torch.float16, device=partials.device) for i in range(nseg)]
    if reorder_index is not None:
        assert reorder_index.dtype == torch.int16 and reorder_index.numel() == k and reorder_index.is_cuda
    if add is not None:
        _require_cuda_half(add, "add")
        assert add.shape == (m, n) and add.is_contiguous()
    st = L.lib().atom_gemm_w4a4_multi_merge_q(partials.data_ptr(), int(splits), L.ptr(reorder_index), float(clip), fused["b4"].data_ptr(),
                                              fused["sb"].data_ptr(), fused["b8"].data_ptr(), fused["sb8"].data_ptr(), outs[0].data_ptr(),
                                              outs[1].data_ptr() if nseg > 1 else None, outs[2].data_ptr() if nseg > 2 else None,
                                              int(f32_mask), L.ptr(add), m, n, nseg, k, GROUP_SIZE, GROUP_SIZE,
                                              L.current_stream(partials.device))
    L.check(st, "atom_gemm_w4a4_multi_merge_q")
    return tuple(outs)


def quant_weight_w4(weight: torch.Tensor, w_clip: float = 0.85, channel_group: int = 2, return_fake_quant=False):
    """NEW (no reference counterpart): quantise + pack a (column-reordered) FP16 weight [N,K] the
    way QLinearLayer.quant does (qLinearLayer.py:42-78).  Returns (B4 u8[N,K4/2], B8 i8[N,128], sB f16[G,N],
    sB8 f16[N][, Wq f16[N,K]])."""
    _require_cuda_half(weight, "weight")
    n, k = weight.shape
    dev = weight.device
    b4 = torch.empty((n, (k - GROUP_SIZE) // 2), dtype=torch.uint8, device=dev)
    b8 = torch.empty((n, GROUP_SIZE), dtype=torch.int8, device=dev)
    sb = torch.empty(((k - GROUP_SIZE) // GROUP_SIZE, n), dtype=torch.float16, device=dev)
    sb8 = torch.empty((n,), dtype=torch.float16, device=dev)
    wq = torch.empty_like(weight) if return_fake_quant else None
    st = L.lib().atom_quant_weight_w4(weight.data_ptr(), n, k, float(w_clip), int(channel_group), b4.data_ptr(),
                                       b8.data_ptr(), sb.data_ptr(), sb8.data_ptr(), L.ptr(wq),
                                       L.current_stream(dev))
    L.check(st, "atom_quant_weight_w4")
    return (b4, b8, sb, sb8, wq) if return_fake_quant else (b4, b8, sb, sb8)


def pack_weight_w4(weight_fq: torch.Tensor, channel_group: int = 2, strict: bool = True):
    """NEW: pack an ALREADY fake-quantised FP16 weight [N,K] -- the tensor GPTQ assigns to
    ``layer.weight.data`` (gptq.py:331) or ``QLinearLayer.quant`` leaves behind -- by recovering codes and fp16 scales.
    Returns (B4, B8, sB, sB8, bad) with ``bad`` = number of (channel_group x 128) blocks that are on no INT4/INT8 grid;
    ``strict`` raises when bad != 0 (costs one device->host sync; this is an offline op)."""
    _require_cuda_half(weight_fq, "weight_fq")
    n, k = weight_fq.shape
    dev = weight_fq.device
    w = weight_fq.contiguous()
    b4 = torch.empty((n, (k - GROUP_SIZE) // 2), dtype=torch.uint8, device=dev)
    b8 = torch.empty((n, GROUP_SIZE), dtype=torch.int8, device=dev)
    sb = torch.empty(((k - GROUP_SIZE) // GROUP_SIZE, n), dtype=torch.float16, device=dev)
    sb8 = torch.empty((n,), dtype=torch.float16, device=dev)
    bad = torch.empty((1,), dtype=torch.int32, device=dev)
    st = L.lib().atom_pack_weight_w4(w.data_ptr(), n, k, int(channel_group), b4.data_ptr(), b8.data_ptr(),
                                      sb.data_ptr(), sb8.data_ptr(), bad.data_ptr(), L.current_stream(dev))
    L.check(st, "atom_pack_weight_w4")
    nbad = int(bad.item())
    if strict and nbad:
        raise L.AtomHipError(f"pack_weight_w4: {nbad} weight blocks are not INT4-g128/INT8 fake-quantised values")
    return b4, b8, sb, sb8, nbad


# ------------------------------------------------------------------------------------------------ INT4 paged KV cache
def _kv_dims(kv):
    _c, num_layers, _2, num_heads, page_size, half_dim = kv.data.shape
    return num_layers, num_heads, page_size, half_dim * 2


def _kv_append(kv, k, v, k_param, v_param, append_indptr, layer_idx, what):
    for t in (kv.data, kv.param, k, v, k_param, v_param):
        if not t.is_cuda:
            raise L.AtomHipError("KV-cache operands must live on the GPU: no CPU fallback")
    num_layers, num_heads, page_size, head_dim = _kv_dims(kv)
    total = k.size(0)
    assert k.shape == v.shape == (total, num_heads, head_dim // 2) and k.dtype == v.dtype == torch.uint8
    assert k_param.numel() == v_param.numel() == total * num_heads * 2 and k_param.dtype == torch.float16
    batch = kv.last_page_offset.numel()
    st = L.lib().atom_kv_append_i4(kv.data.data_ptr(), kv.param.data_ptr(), kv.indptr.data_ptr(),
                                    kv.indicies.data_ptr(), kv.last_page_offset.data_ptr(), k.contiguous().data_ptr(),
                                    v.contiguous().data_ptr(), k_param.contiguous().data_ptr(),
                                    v_param.contiguous().data_ptr(), L.ptr(append_indptr), total, batch, num_layers,
                                    int(layer_idx), num_heads, page_size, head_dim, L.current_stream(k.device))
    L.check(st, what)


def init_kv_i4(kv, k, v, k_param, v_param, seqlen_indptr, layer_idx: int):
    """Write the prefill tokens' quantised K/V into the pages.  Reference: punica/ops/__init__.py:33-45 ->
    FlashInferInitKvKernel_i4.  k, v uint8 [sum(len), heads, 64]; k_param, v_param fp16 [sum(len), heads, 2];
    seqlen_indptr int32 [batch+1]."""
    assert seqlen_indptr.dtype == torch.int32 and seqlen_indptr.is_cuda
    _kv_append(kv, k, v, k_param, v_param, seqlen_indptr, layer_idx, "atom_kv_append_i4 (init)")


def append_kv_i4(kv, k, v, k_param, v_param, layer_idx: int):
    """Append ONE token per sequence.  Reference: punica/ops/__init__.py:48-59 -> FlashInferAppendKvKernel_i4."""
    _kv_append(kv, k, v, k_param, v_param, None, layer_idx, "atom_kv_append_i4")


def quant_append_kv_i4(kv, k_f32: torch.Tensor, v_f32: torch.Tensor, layer_idx: int):
    """NEW (fused decode step): quantise the FP32 k / v projections [batch, heads*128] per head (the _o4 epilogue) and
    append them as the last token of every sequence.  Same cache contents as dense_layer_gemm_i4_o4 + append_kv_i4."""
    num_layers, num_heads, page_size, head_dim = _kv_dims(kv)
    batch = kv.last_page_offset.numel()
    for t in (k_f32, v_f32):
        if not t.is_cuda:
            raise L.AtomHipError("KV-cache operands must live on the GPU: no CPU fallback")
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == (batch, num_heads * head_dim)
    st = L.lib().atom_kv_quant_append_f32(kv.data.data_ptr(), kv.param.data_ptr(), kv.indptr.data_ptr(),
                                          kv.indicies.data_ptr(), kv.last_page_offset.data_ptr(), k_f32.data_ptr(),
                                          v_f32.data_ptr(), batch, num_layers, int(layer_idx), num_heads, page_size, head_dim,
                                          L.current_stream(k_f32.device))
    L.check(st, "atom_kv_quant_append_f32")


def decode_splits(batch: int, kv) -> int:
    """How many partial states per (sequence, head) batch_decode_i4 produces for this cache (1: none; small batches: one per workgroup of
    four KV-split waves)."""
    num_layers, num_heads, page_size, head_dim = _kv_dims(kv)
    return int(L.lib().atom_batch_decode_i4_splits(int(batch), num_heads, page_size, int(getattr(kv, "max_pages", 0))))


def batch_decode_i4(q: torch.Tensor, kv, layer_idx: int, *, rope_theta: float = 1e4, rope_scale: float = 1.0, append_kv=None, merge=True):
    """Decode attention over the INT4 paged cache, RoPE fused.  Reference: punica/ops/__init__.py:21-30 ->
    FlashInferBatchDecodeKernel_i4 (rope_theta 1e4, rope_scale 1 hard-coded there).  q fp16 [batch, heads, 128].
    ``append_kv=(k_f32, v_f32)`` (round 6): quant_append_kv_i4 of this step's FP32 k / v projections inside the same launch
    (atom_batch_decode_append_i4) -- same cache contents, same output, one launch fewer.
    ``merge=False`` (only where decode_splits(batch, kv) >= 2): no merge launch -- returns the split partial states float32
    [batch, heads, splits, 130] (a tensor of their own, not the shared workspace) for dense_layer_gemm_i4_merge_q."""
    _require_cuda_half(q, "q")
    num_layers, num_heads, page_size, head_dim = _kv_dims(kv)
    batch = q.size(0)
    assert q.shape == (batch, num_heads, head_dim)
    lib = L.lib()
    max_pages = int(getattr(kv, "max_pages", 0))
    ws_bytes = lib.atom_batch_decode_i4_workspace_bytes(batch, num_heads, page_size, max_pages)
    if not merge:
        splits = lib.atom_batch_decode_i4_splits(batch, num_heads, page_size, max_pages)
        assert splits >= 2 and ws_bytes == batch * num_heads * splits * 130 * 4, "merge=False needs a split KV range (decode_splits)"
        part = torch.empty((batch, num_heads, splits, 130), dtype=torch.float32, device=q.device)
        o, ws = None, part
    else:
        o = torch.empty_like(q)
        ws = _workspace(q.device, ws_bytes) if ws_bytes else None
    if append_kv is not None:
        k32, v32 = append_kv
        for t in (k32, v32):
            if not t.is_cuda:
                raise L.AtomHipError("KV-cache operands must live on the GPU: no CPU fallback")
            assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == (batch, num_heads * head_dim)
        assert batch == kv.last_page_offset.numel()
        st = lib.atom_batch_decode_append_i4(L.ptr(o), q.data_ptr(), k32.data_ptr(), v32.data_ptr(), kv.data.data_ptr(),
                                             kv.param.data_ptr(), kv.indptr.data_ptr(), kv.indicies.data_ptr(),
                                             kv.last_page_offset.data_ptr(), batch, num_layers, int(layer_idx), num_heads, page_size,
                                             head_dim, float(rope_theta), float(rope_scale), max_pages, L.ptr(ws), ws_bytes,
                                             L.current_stream(q.device))
        L.check(st, "atom_batch_decode_append_i4")
        return o if merge else part
    st = lib.atom_batch_decode_i4(L.ptr(o), q.data_ptr(), kv.data.data_ptr(), kv.param.data_ptr(),
                                  kv.indptr.data_ptr(), kv.indicies.data_ptr(), kv.last_page_offset.data_ptr(), batch,
                                  num_layers, int(layer_idx), num_heads, page_size, head_dim, float(rope_theta),
                                  float(rope_scale), max_pages, L.ptr(ws), ws_bytes, L.current_stream(q.device))
    L.check(st, "atom_batch_decode_i4")
    return o if merge else part


def kv_fake_quant(x: torch.Tensor, n_bits: int = 4, clip: float = 1.0) -> torch.Tensor:
    """Asymmetric per-head-vector fake quantisation of a [batch, heads, seq, 128] fp16 tensor (any strides over the first
    three dims) -- quantize_attn_k_wrapper / quantize_attn_v_wrapper of the reference (model/quant.py:233-257)."""
    if not x.is_cuda:
        raise L.AtomHipError("kv_fake_quant needs a GPU tensor: no CPU fallback")
    assert x.dtype == torch.float16 and x.dim() == 4 and x.shape[-1] == 128 and x.stride(-1) == 1
    b, h, s, _ = x.shape
    y = torch.empty((b, h, s, 128), dtype=torch.float16, device=x.device)
    st = L.lib().atom_kv_fake_quant_f16(x.data_ptr(), y.data_ptr(), b, h, s, x.stride(0), x.stride(1), x.stride(2),
                                         int(n_bits), float(clip), L.current_stream(x.device))
    L.check(st, "atom_kv_fake_quant_f16")
    return y


def scale_pairs_shared(b_scale: torch.Tensor, n: int) -> bool:
    """True when output channels 2j and 2j+1 share their weight scale in every group (weight_channel_group = 2 -- the only form the
    reference kernel accepts, Dense_layer_gemm_i4_o16.cuh:413-431): the GEMM is then told so (ATOM_B_SCALE_PAIRS) and forms each scale
    product once per pair.  Checked on the values (one device round trip, offline with the weight; never during graph capture)."""
    if n % 2 or torch.cuda.is_current_stream_capturing():
        return False
    s = b_scale.reshape(-1, n)
    return bool((s[:, 0::2] == s[:, 1::2]).all().item())


def _pairs_flag(b_scale: torch.Tensor, n: int) -> bool:
    """scale_pairs_shared() of a weight-scale tensor, remembered ON the tensor object together with its version counter: one device
    round trip per weight (and per in-place rewrite), none for a model's parameters afterwards -- pass the SAME tensor object every
    time (a Parameter, not a fresh ``.data`` view: the tag would die with the view and every call would pay the round trip).  False
    during graph capture for a tensor not seen before (the generic kernels are always right)."""
    tag = getattr(b_scale, "_atom_pairs", None)
    ver = _version_of(b_scale)                # (inference tensors: -1 -- they cannot be written in place)
    if tag is not None and tag[0] == ver:
        return tag[1]
    if torch.cuda.is_current_stream_capturing():
        return False
    v = scale_pairs_shared(b_scale, n)
    try:
        b_scale._atom_pairs = (ver, v)
    except AttributeError:
        pass
    return v


def repack_weight_f6(b4: torch.Tensor, b_scale: torch.Tensor = None) -> torch.Tensor:
    """Packed INT4 weights uint8 [N, K4/2] -> the F6 operand format uint8 [G][f6_rows(N)][104] (atom_repack_weight_f6).
    With the fp16 weight scales `b_scale` [G, N]: atom_repack_weight_f6s -- the returned tensor is a view of a buffer that
    carries the scales as float32 [G][f6_rows(N)] behind the codes (ATOM_B_F6S; dense_layer_gemm_i4_fp16 recognises it by the
    `atom_f6s` attribute and the 256x256 kernel then takes its weight scales from there)."""
    if not b4.is_cuda:
        raise L.AtomHipError("repack_weight_f6 needs a GPU tensor: no CPU fallback")
    n, k4h = b4.shape
    g = k4h // 64
    if b_scale is not None:
        assert b_scale.is_cuda and b_scale.dtype == torch.float16 and b_scale.numel() == g * n
        nbytes = L.lib().atom_f6_weight_bytes(n, k4h * 2 + GROUP_SIZE)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=b4.device)
        st = L.lib().atom_repack_weight_f6s(b4.contiguous().data_ptr(), b_scale.contiguous().data_ptr(), n, k4h * 2 + GROUP_SIZE,
                                             buf.data_ptr(), L.current_stream(b4.device))
        L.check(st, "atom_repack_weight_f6s")
        out = buf[:g * f6_rows(n) * L.F6_PITCH].view(g, f6_rows(n), L.F6_PITCH)
        out.atom_f6s = buf                       # the whole buffer (codes + float32 scales) stays alive with the view
        out.atom_pairs = scale_pairs_shared(b_scale, n)
        return out
    out = torch.empty((g, f6_rows(n), L.F6_PITCH), dtype=torch.uint8, device=b4.device)
    st = L.lib().atom_repack_weight_f6(b4.contiguous().data_ptr(), n, k4h * 2 + GROUP_SIZE, out.data_ptr(),
                                        L.current_stream(b4.device))
    L.check(st, "atom_repack_weight_f6")
    return out


def repack_act_f6(a4: torch.Tensor, a_scale: torch.Tensor, *, scale_layout="ref") -> torch.Tensor:
    """Packed INT4 activations [M, K4/2] + scales -> the F6 activation operand [G][f6_rows(M)][104] (atom_repack_act_f6)."""
    if not a4.is_cuda:
        raise L.AtomHipError("repack_act_f6 needs a GPU tensor: no CPU fallback")
    m, k4h = a4.shape
    out = torch.empty((k4h // 64, f6_rows(m), L.F6_PITCH), dtype=torch.uint8, device=a4.device)
    st = L.lib().atom_repack_act_f6(a4.data_ptr(), a_scale.data_ptr(), m, k4h * 2 + GROUP_SIZE, _LAYOUTS[scale_layout], out.data_ptr(),
                                    L.current_stream(a4.device))
    L.check(st, "atom_repack_act_f6")
    return out


# ----------------------------------------------------------------------------------------------- fused gate / up
def fuse_gate_up_weights(gate, up):
    """(b4 u8 [N, K4/2], b8 i8 [N, 128], sb f16 [G, N], sb8 f16 [N]) of gate_proj and of up_proj -> the fused weight operand of
    gate_up_silu_quant_f6: rows interleaved per 128-feature block and 32-feature quarter (32 gate rows, then the same 32 up rows),
    codes + float32 scales in the F6 format (atom_repack_weight_f6s).  Offline, once per layer.  Returns a dict."""
    b4g, b8g, sbg, sb8g = gate
    b4u, b8u, sbu, sb8u = up
    n = b4g.shape[0]
    assert b4u.shape == b4g.shape and n % GROUP_SIZE == 0 and n >= 2 * GROUP_SIZE
    dev = b4g.device
    q = torch.arange(n, device=dev).view(n // 128, 4, 32)                                   # [block, quarter, 32 features]
    idx = torch.stack([q, q + n], dim=2).reshape(-1)                                        # gate rows, then the same up rows
    b4 = torch.cat([b4g.view(torch.uint8), b4u.view(torch.uint8)], 0).index_select(0, idx).contiguous()
    b8 = torch.cat([b8g, b8u], 0).index_select(0, idx).contiguous()
    g = sbg.numel() // n
    sb = torch.cat([sbg.reshape(g, n), sbu.reshape(g, n)], 1).index_select(1, idx).contiguous()
    sb8 = torch.cat([sb8g.reshape(-1), sb8u.reshape(-1)], 0).index_select(0, idx).contiguous()
    return {"b6s": repack_weight_f6(b4, sb), "b8": b8, "sb8": sb8, "n_inter": n, "k": b4g.shape[1] * 2 + GROUP_SIZE}


def gate_up_silu_quant_f6(a6, a_keeper, a_keeper_scale, fused, *, quant_mode="kernel", clip=1.0, scale_layout="ref",
                          return_dequant=False):
    """quant(silu(x Wg^T) * (x Wu^T)) in one launch, from the F6 activation operand of x straight to the F6 activation operand of
    down_proj: the reference's gate_proj, up_proj and activate_fp16_i4 (punica/models/llama.py:85-87) -- same return tuple as
    activate_fp16_i4(..., wide_codes="f6"), bit-identical to it on the fp16 GEMM outputs."""
    m = a_keeper.size(0)
    n, k = fused["n_inter"], fused["k"]
    assert a6.shape == (k // GROUP_SIZE - 1, f6_rows(m), L.F6_PITCH)
    outs = _alloc_act_outputs(m, n, a6.device, scale_layout, return_dequant, "f6")
    b6s = fused["b6s"]
    st = L.lib().atom_gemm_w4a4_silu_mul_quant_f6(a6.data_ptr(), b6s.atom_f6s.data_ptr(), a_keeper.data_ptr(), fused["b8"].data_ptr(),
                                                  a_keeper_scale.data_ptr(), fused["sb8"].data_ptr(), m, n, k, GROUP_SIZE, GROUP_SIZE,
                                                  _MODES[quant_mode], float(clip),
                                                  _LAYOUTS[scale_layout] | (L.B_SCALE_PAIRS if getattr(b6s, "atom_pairs", False) else 0),
                                                  outs[0].data_ptr(),
                                                  outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), L.ptr(outs[4]),
                                                  L.current_stream(a6.device))
    L.check(st, "atom_gemm_w4a4_silu_mul_quant_f6")
    return _ret(*outs)
