"""CPU oracle for Atom's W4A4 mixed-precision GEMM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``atom_amd/``) may import
this module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker.

This is a numpy restatement of the reference algorithms (file:line relative to
/root/reference):

* simulated ("fake") quantisation -- ``model/quant.py:118-183`` (quantize_tensor),
  ``:68-107`` (quantize_tensor_channel_group), ``:187-231``
  (quantize_activation_wrapper), ``model/qLinearLayer.py:42-78`` (QLinearLayer.quant).
  Restated in the *integer domain*: we return (codes, scales) such that
  ``codes * scales`` reproduces the reference's FP16 fake-quant tensor bit for bit.
  PINNED: ``tests/golden/gen_golden.py`` runs the unmodified reference modules and
  stores their outputs; ``tests/test_oracle_golden.py`` checks this file against them.
* kernel-flavoured quantisation (no clip, FP32 ``x*(1/s)``, round-half-away) --
  ``kernels/include/Reorder/Reorder.cuh:64-190``, ``RMSNorm/RMSNorm.cuh:66-238``,
  ``Activate/Activate.cuh:67-180`` and their CPU goldens
  ``e2e/punica-atom/punica/ops/csrc/Reorder/test_Reorder.cu:41-112`` etc.
  The reference holds no stored vectors for these, but its unit tests carry CPU golden
  FUNCTIONS: PINNED against those, compiled from /root/reference into oracle/_ref
  (oracle/Makefile, oracle/ref/), in tests/test_oracle_ref.py -- scales bit-equal (one fp16
  ulp where expf / the FP32-vs-half normalisation enters), codes +-1 on a stated fraction
  (x * (1/s) vs x / s at rounding ties), tighter than the reference's own acceptance.
* the GEMM -- ``kernels/include/GEMM/Dense_layer_gemm_i4_o16.cuh:404-434`` (dequant),
  ``:713-727`` (operand doc), A-scale layout ``Reorder.cuh:39-50``.
  GEMM numerics are pinned by NO reference test ("parity unpinned"): the
  golden for it is constructed from the reference's simulated path
  (``F.linear`` on fake-quant operands, ``model/qLinearLayer.py:32-35``).

All FP16 arithmetic follows torch's CPU/GPU "opmath" convention, which numpy's
float16 also uses: operands widened to float32, one float32 op, result rounded to
float16 (round-to-nearest-even).
"""
from __future__ import annotations

import numpy as np

GROUP = 128
KEEPER = 128

f16 = np.float16
f32 = np.float32


# --------------------------------------------------------------------------- layout helpers
def scale_index(row: int) -> int:
    """Offset (in halves) of row's first scale replica. Reorder.cuh:39-44."""
    bottom_upper = (row // 8) % 2
    group_idx = row % 8
    group_nums = row // 16
    return group_nums * 64 + group_idx * 8 + bottom_upper


def scale_size(rows: int) -> int:
    """Leading dimension (halves) of the replicated A-scale layout.
    SCALE_SIZE_A, Reorder.cuh:50 == punica/ops/__init__.py:137-138."""
    x = rows
    return x // 16 * 64 + 64 - (1 - (x % 16) // 8) * (8 - (x % 8)) * 8


def scales_to_ref_layout(s: np.ndarray) -> np.ndarray:
    """[G, M] plain -> [G, scale_size(M)] replicated layout (4 replicas at +0,+2,+4,+6).
    Reorder.cuh:137-156.  Unwritten slots are zero."""
    s = np.asarray(s, dtype=f16)
    one_d = s.ndim == 1
    if one_d:
        s = s[None, :]
    G, M = s.shape
    out = np.zeros((G, scale_size(M)), dtype=f16)
    for r in range(M):
        b = scale_index(r)
        for j in range(4):
            out[:, b + 2 * j] = s[:, r]
    return out[0] if one_d else out


def scales_from_ref_layout(s: np.ndarray, M: int) -> np.ndarray:
    s = np.asarray(s)
    one_d = s.ndim == 1
    if one_d:
        s = s[None, :]
    idx = np.array([scale_index(r) for r in range(M)], dtype=np.int64)
    out = s[:, idx]
    return out[0] if one_d else out


def pack_int4(codes: np.ndarray) -> np.ndarray:
    """int8 codes in [-8,7], last dim even -> uint8, element 2j in the LOW nibble,
    2j+1 in the HIGH nibble (two's complement). PackInt4, Reorder.cuh:16-19,174-176."""
    c = np.asarray(codes).astype(np.int16)
    lo = c[..., 0::2] & 0xF
    hi = c[..., 1::2] & 0xF
    return (lo | (hi << 4)).astype(np.uint8)


def unpack_int4(packed: np.ndarray) -> np.ndarray:
    p = np.asarray(packed, dtype=np.uint8).astype(np.int16)
    lo = p & 0xF
    hi = (p >> 4) & 0xF
    lo = np.where(lo >= 8, lo - 16, lo)
    hi = np.where(hi >= 8, hi - 16, hi)
    out = np.empty(p.shape[:-1] + (p.shape[-1] * 2,), dtype=np.int8)
    out[..., 0::2] = lo
    out[..., 1::2] = hi
    return out


# --------------------------------------------------------------------------- rounding helpers
def _rne(x):
    return np.rint(x)  # round half to even == torch.round


def _round_half_away(x):
    """C/CUDA round(): half away from zero (Reorder.cuh:171-172)."""
    x = np.asarray(x, dtype=f32)
    t = np.trunc(x)
    return (t + np.sign(x) * (np.abs(x - t) >= f32(0.5))).astype(f32)


# --------------------------------------------------------------------------- quantisation tails
def quant_groups_sim(v16: np.ndarray, n_bits: int, clip: float):
    """quantize_tensor(sym=True, group_size=0) on rows of ``v16`` -- quant.py:141-142,166-172,181.

    v16: float16 [rows, group].  Returns (codes int8 [rows, group], scale float16 [rows]).
    Every step is one float32 op rounded to float16, exactly as torch does for half tensors.
    """
    v16 = np.asarray(v16, dtype=f16)
    qmax = 2 ** (n_bits - 1) - 1
    qmin = -(2 ** (n_bits - 1))
    amax = np.abs(v16).max(axis=-1)                        # exact in fp16
    amax = np.maximum(amax, f16(1e-5))                     # .clamp(min=1e-5): scalar cast to half
    if clip < 1.0:
        amax = (amax.astype(f32) * f32(clip)).astype(f16)  # w_max * clip_ratio
    scale = (amax.astype(f32) / f32(qmax)).astype(f16)     # scales = w_max / q_max
    q = (v16.astype(f32) / scale.astype(f32)[..., None]).astype(f16)   # w / scales (half)
    q = np.clip(_rne(q.astype(f32)), qmin, qmax)           # clamp(round(.))
    return q.astype(np.int8), scale


def dequant_sim(codes: np.ndarray, scale: np.ndarray) -> np.ndarray:
    """(clamp(round(w/s)) - 0) * s in half -- quant.py:181."""
    with np.errstate(over="ignore"):
        return (codes.astype(f32) * scale.astype(f32)[..., None]).astype(f16)


def quant_groups_kernel(v32: np.ndarray, n_bits: int, clip: float = 1.0):
    """The CUDA kernels' tail -- Reorder.cuh:137-178 (== RMSNorm.cuh:153-237, Activate.cuh:112-167).

    v32: float32 [rows, group] (values already rounded to fp16 where the kernel does so).
    scale_f = amax / qmax (fp32); stored scale = half(scale_f); r = 1/scale_f;
    q = clamp(round_half_away(v * r)).
    Divergence, documented: an all-zero group gives scale 0 and codes 0 here (the reference
    computes 0 * inf = NaN -> undefined int cast).
    """
    v32 = np.asarray(v32, dtype=f32)
    qmax = 2 ** (n_bits - 1) - 1
    qmin = -(2 ** (n_bits - 1))
    amax = np.abs(v32).max(axis=-1).astype(f32)
    if clip < 1.0:
        amax = (amax * f32(clip)).astype(f32)
    scale_f = (amax / f32(qmax)).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (f32(1.0) / scale_f).astype(f32)
        q = _round_half_away((v32 * r[..., None]).astype(f32))
    q = np.where(scale_f[..., None] == 0, f32(0), q)
    q = np.clip(q, qmin, qmax)
    return q.astype(np.int8), scale_f.astype(f16)


def _quant_row_tail(y, mode: str, clip: float):
    """Shared tail of the three activation ops: split [M,K] into G INT4 groups + 1 INT8 keeper group.

    y: float16 (sim) or float32 (kernel) [M, K], already in reordered channel order.
    Returns dict(q4 int8 [M,K4], s4 f16 [M,G], q8 int8 [M,128], s8 f16 [M]).
    """
    M, K = y.shape
    assert K % GROUP == 0 and K > KEEPER
    K4 = K - KEEPER
    G = K4 // GROUP
    body = y[:, :K4].reshape(M * G, GROUP)
    keep = y[:, K4:]
    if mode == "sim":
        q4, s4 = quant_groups_sim(body, 4, clip)           # quant.py:192-201,225
        q8, s8 = quant_groups_sim(keep, 8, 1.0)            # quant.py:219-220 (no clip)
    elif mode == "kernel":
        q4, s4 = quant_groups_kernel(body, 4, clip)
        q8, s8 = quant_groups_kernel(keep, 8, 1.0)
    else:
        raise ValueError(mode)
    return dict(q4=q4.reshape(M, K4), s4=s4.reshape(M, G), q8=q8, s8=s8)


def act_dequant_sim(t) -> np.ndarray:
    """Rebuild the FP16 tensor quantize_activation_wrapper returns (quant.py:225-228)."""
    M, K4 = t["q4"].shape
    G = K4 // GROUP
    body = dequant_sim(t["q4"].reshape(M * G, GROUP), t["s4"].reshape(M * G)).reshape(M, K4)
    keep = dequant_sim(t["q8"], t["s8"])
    return np.concatenate([body, keep], axis=1)


# --------------------------------------------------------------------------- the three activation ops
def reorder_quant(x16: np.ndarray, reorder_index, mode: str = "sim", clip: float = 0.9):
    """index_select(x, -1, idx) then dynamic per-token quant.
    sim: qLlamaLayer.py:300-304 + quant.py:187-231.  kernel: Reorder.cuh:64-190."""
    x16 = np.asarray(x16, dtype=f16)
    idx = np.asarray(reorder_index).astype(np.int64)
    y = x16[:, idx]
    if mode == "kernel":
        y = y.astype(f32)
    return _quant_row_tail(y, mode, clip)


def sumsq_tree(x16) -> np.ndarray:
    """Sum of squares of each row as the HIP RMSNorm kernels take it (quant_kernels.hip): a fixed-shape FP32 tree over the row in
    memory order.  16-byte chunk c (8 halves) belongs to thread (wave (c // 64) % 4, lane c % 64); a thread folds its chunks in
    order with s = fma(x, x, s); the 64 lanes of a wave combine by the butterfly xor 32, 16, 8, 4, 2, 1; the four waves as
    ((w0 + w1) + w2) + w3.  (fma: exact product and sum in float64, rounded to float32 once.)"""
    x = np.asarray(x16, dtype=f16).astype(np.float64)
    M, H = x.shape
    assert H % 8 == 0
    it = -(-(H // 8) // 256)
    xp = np.zeros((M, it * 256 * 8), dtype=np.float64)
    xp[:, :H] = x
    v = xp.reshape(M, it, 4, 64, 8)
    s = np.zeros((M, 4, 64), dtype=f32)
    for i in range(it):
        for k in range(8):
            s = (s.astype(np.float64) + v[:, i, :, :, k] ** 2).astype(f32)
    lanes = np.arange(64)
    for m in (32, 16, 8, 4, 2, 1):
        s = (s + s[:, :, lanes ^ m]).astype(f32)
    w = s[:, :, 0]
    return (((w[:, 0] + w[:, 1]).astype(f32) + w[:, 2]).astype(f32) + w[:, 3]).astype(f32)


def rmsnorm_f16(x16, w16, eps: float, mode: str) -> np.ndarray:
    """RMSNorm producing the FP16 tensor that is then quantised.

    sim   : HF LlamaRMSNorm (transformers 4.39 modeling_llama.py, called at qLlamaLayer.py:143):
            var = mean(float(x)^2); y = half(float(x) * rsqrt(var+eps)); out = half(w * y).
    kernel: RMSNorm.cuh:112-151: out = half(float(x) * float(w) * r).
    Sum of squares: the fixed-shape FP32 tree of sumsq_tree() (any order is within the reference golden's tolerance -- the
    reference itself sums with torch's reduction order; this one is specified so that the HIP kernel can be followed to the bit);
    var = ss / H and r = 1/sqrt(var+eps) with correctly rounded fp32 divide and sqrt.
    """
    x16 = np.asarray(x16, dtype=f16)
    w16 = np.asarray(w16, dtype=f16)
    H = x16.shape[-1]
    var = (sumsq_tree(x16.reshape(-1, H)).reshape(x16.shape[:-1]) / f32(H)).astype(f32)
    r = (f32(1.0) / np.sqrt((var + f32(eps)).astype(f32))).astype(f32)
    xf = x16.astype(f32)
    if mode == "sim":
        y = (xf * r[:, None]).astype(f16)
        return (w16.astype(f32)[None, :] * y.astype(f32)).astype(f16)
    t = (xf * w16.astype(f32)[None, :]).astype(f32)
    return (t * r[:, None]).astype(f16)


def rmsnorm_reorder_quant(x16, w16, eps, reorder_index, mode: str = "sim", clip: float = 0.9):
    """sim: qLlamaLayer.py:141-151.  kernel: RMSNorm.cuh:66-238."""
    y = rmsnorm_f16(x16, w16, eps, mode)
    idx = np.asarray(reorder_index).astype(np.int64)
    y = y[:, idx]
    if mode == "kernel":
        y = y.astype(f32)
    return _quant_row_tail(y, mode, clip)


def silu_mul(a16, b16, mode: str):
    """sim: act_fn(gate)*up in half (qLlamaLayer.py:346-348): half(half(silu(a)) * b).
    kernel: silu(float(a))*float(b) kept in FP32 (Activate.cuh:28,103-106)."""
    a = np.asarray(a16, dtype=f16).astype(f32)
    b = np.asarray(b16, dtype=f16).astype(f32)
    s = (a / (f32(1.0) + np.exp(-a).astype(f32))).astype(f32)
    if mode == "sim":
        s = s.astype(f16).astype(f32)
        return (s * b).astype(f16)
    return (s * b).astype(f32)


def silu_mul_quant(a16, b16, mode: str = "sim", clip: float = 0.9):
    """sim: qLlamaLayer.py:345-351.  kernel: Activate.cuh:67-180. No reorder (weights pre-permuted)."""
    return _quant_row_tail(silu_mul(a16, b16, mode), mode, clip)


# --------------------------------------------------------------------------- weights
def quant_weight_sim(W16: np.ndarray, w_clip: float = 0.85, channel_group: int = 2):
    """QLinearLayer.quant (qLinearLayer.py:42-78) in the integer domain.

    W16 [N,K] float16 (already column-reordered).  Keeper = last 128 columns -> INT8 per output row,
    no clip (:58).  Remaining columns: for each 128-column slice, ``channel_group`` adjacent rows share
    one scale (quant.py:86-87), clip w_clip, 4 bit.
    Returns dict(q4 int8 [N,K4], s4 f16 [G,N], q8 int8 [N,128], s8 f16 [N], wq f16 [N,K]).
    """
    W16 = np.asarray(W16, dtype=f16)
    N, K = W16.shape
    K4 = K - KEEPER
    G = K4 // GROUP
    cg = channel_group
    assert N % cg == 0
    q8, s8 = quant_groups_sim(W16[:, K4:], 8, 1.0)
    q4 = np.empty((N, K4), dtype=np.int8)
    s4 = np.empty((G, N), dtype=f16)
    for g in range(G):
        blk = W16[:, g * GROUP:(g + 1) * GROUP].reshape(N // cg, cg * GROUP)
        q, s = quant_groups_sim(blk, 4, w_clip)
        q4[:, g * GROUP:(g + 1) * GROUP] = q.reshape(N, GROUP)
        s4[g] = np.repeat(s, cg)
    wq = np.empty((N, K), dtype=f16)
    for g in range(G):
        wq[:, g * GROUP:(g + 1) * GROUP] = dequant_sim(q4[:, g * GROUP:(g + 1) * GROUP], s4[g])
    wq[:, K4:] = dequant_sim(q8, s8)
    return dict(q4=q4, s4=s4, q8=q8, s8=s8, wq=wq)


def gptq_codes(q16: np.ndarray, s32: np.ndarray, n_bits: int, channel_group: int):
    """Codes the reference GPTQ wrote: quantize_gptq (gptq.py:38-39) computes ``scale*(clamp(round(x/scale)+zero,0,maxq)
    -zero)`` in FP32 with zero=(maxq+1)/2 (sym, gptq.py:139-140); the stored weight is half() of that (gptq.py:331), so
    code = round(float(q16)/scale).  ``q16`` [N,128] one group, ``s32`` [N/channel_group] the scale find_params produced
    for it (gptq.py:285-287)."""
    N = q16.shape[0]
    sc = np.repeat(np.asarray(s32, dtype=np.float32).reshape(-1), channel_group).reshape(N, 1)
    c = np.rint(q16.astype(np.float32) / sc)
    lo, hi = -(1 << (n_bits - 1)), (1 << (n_bits - 1)) - 1
    assert c.min() >= lo and c.max() <= hi
    return c.astype(np.int8)


def _recover_blocks(v: np.ndarray, n_bits: int):
    """Recover (codes, fp16 scale) of blocks whose values are half(s*c) -- CPU restatement of weight_pack_kernel
    (atom_amd/csrc/quant_kernels.hip), the packer for what gptq.py:331 / qLinearLayer.py:42-78 leave in ``weight``.
    v [B, L] float32 (exact fp16 values).  Returns codes int8 [B,L], scale f16 [B], bad bool [B]."""
    v = np.asarray(v, dtype=np.float32)
    B, L = v.shape
    qmax, qmin = float((1 << (n_bits - 1)) - 1), float(-(1 << (n_bits - 1)))
    kmax = 1 << (n_bits - 1)
    tol = np.abs(v) * np.float32(2.0 ** -9) + np.float32(2.0 ** -24)
    amax = np.abs(v).max(axis=1)
    best_err = np.full(B, np.inf, dtype=np.float32)
    best_s = np.zeros(B, dtype=np.float32)
    best_q = np.zeros((B, L), dtype=np.float32)
    done = amax == 0
    best_err[done] = 0
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for k in range(kmax, 0, -1):
            live = ~done
            if not live.any():
                break
            s = (amax / np.float32(k)).astype(np.float32)[:, None]
            t = np.rint(v / s)
            ok = live & np.all((t >= qmin) & (t <= qmax) & (np.abs(t * s - v) <= tol), axis=1)
            if not ok.any():
                continue
            q = np.clip(t, qmin, qmax)
            den = (q * q).sum(axis=1, dtype=np.float32)
            num = (v * q).sum(axis=1, dtype=np.float32)
            s0 = np.where(den > 0, num / np.where(den > 0, den, 1), s[:, 0]).astype(f16)
            bits0 = s0.view(np.uint16).astype(np.int32)
            for off in (0, -1, 1, -2, 2):
                sc = (bits0 + off).astype(np.uint16).view(f16).astype(np.float32)
                e = (np.abs((q * sc[:, None]).astype(f16).astype(np.float32) - v) / tol).max(axis=1)
                e = np.where((sc > 0) & (sc < 65504), e, np.inf).astype(np.float32)
                upd = ok & (e < best_err)
                best_err[upd] = e[upd]
                best_s[upd] = sc[upd]
                best_q[upd] = q[upd]
            done = done | (ok & (best_err == 0))
    bad = best_err > 1
    if bad.any():
        sb = (np.maximum(amax[bad], np.float32(1e-5)) / np.float32(qmax)).astype(f16).astype(np.float32)
        best_s[bad] = sb
        best_q[bad] = np.clip(np.rint(v[bad] / sb[:, None]), qmin, qmax)
    return best_q.astype(np.int8), best_s.astype(f16), bad


def pack_weight_fq(Wq16: np.ndarray, channel_group: int = 2):
    """atom_pack_weight_w4 on the CPU: same outputs as quant_weight_sim (minus wq) + ``bad`` = number of blocks on no grid."""
    Wq16 = np.asarray(Wq16, dtype=f16)
    N, K = Wq16.shape
    K4 = K - KEEPER
    G = K4 // GROUP
    cg = channel_group
    q8, s8, bad8 = _recover_blocks(Wq16[:, K4:].astype(np.float32), 8)
    q4 = np.empty((N, K4), dtype=np.int8)
    s4 = np.empty((G, N), dtype=f16)
    nbad = int(bad8.sum())
    for g in range(G):
        blk = Wq16[:, g * GROUP:(g + 1) * GROUP].astype(np.float32).reshape(N // cg, cg * GROUP)
        q, s, b = _recover_blocks(blk, 4)
        q4[:, g * GROUP:(g + 1) * GROUP] = q.reshape(N, GROUP)
        s4[g] = np.repeat(s, cg)
        nbad += int(b.sum())
    return dict(q4=q4, s4=s4, q8=q8, s8=s8, bad=nbad)


# --------------------------------------------------------------------------- GEMM
def _group_int_dots(qa4, qb4, g):
    a = qa4[:, g * GROUP:(g + 1) * GROUP].astype(f32)
    b = qb4[:, g * GROUP:(g + 1) * GROUP].astype(f32)
    return a @ b.T  # exact: |sum| <= 128*64 < 2^24


def gemm_w4a4_exact(qa4, qb4, sA, sB, qa8, qb8, sA8, sB8) -> np.ndarray:
    """D = sum_g (A4_g . B4_g^T) sA[m,g] sB[g,n] + (A8 . B8^T) sA8[m] sB8[n] in float64.

    qa4 int8 [M,K4], qb4 int8 [N,K4], sA f16 [M,G], sB f16 [G,N], qa8 int8 [M,128], qb8 int8 [N,128],
    sA8 f16 [M], sB8 f16 [N].  Mathematical value of the reference GEMM
    (Dense_layer_gemm_i4_o16.cuh:404-434, :590-691) with no intermediate rounding."""
    M, K4 = qa4.shape
    G = K4 // GROUP
    D = np.zeros((M, qb4.shape[0]), dtype=np.float64)
    for g in range(G):
        I = _group_int_dots(qa4, qb4, g).astype(np.float64)
        D += I * sA[:, g].astype(np.float64)[:, None] * sB[g].astype(np.float64)[None, :]
    I8 = (qa8.astype(f32) @ qb8.astype(f32).T).astype(np.float64)  # |sum| <= 2^21, exact
    D += I8 * sA8.astype(np.float64)[:, None] * sB8.astype(np.float64)[None, :]
    return D


def gemm_w4a4_ref(qa4, qb4, sA, sB, qa8, qb8, sA8, sB8) -> np.ndarray:
    """The reference kernel's rounding order UP TO FMA CONTRACTION: scale product in FP16 (__hmul2, :417), widened to
    FP32, c_fp32 += float(int_acc) * s per group in group order then keeper (:404-434, :680-691), D = half(c) (:240-243).
    This restatement rounds the product int_acc * s to FP32 and then adds; nvcc contracts `accu += c_frag * rs_scale`
    (:420-431) into one FMA by default, i.e. the compiled reference rounds once per group where this rounds twice --
    immaterial at the 1e-2 this function is used for (the tests' tolerance), and neither is the contract of
    include/atom_hip.h (gemm_w4a4_contract below: exact FP32 scale product, one FMA).  Returns float16 [M,N]."""
    M, K4 = qa4.shape
    G = K4 // GROUP
    c = np.zeros((M, qb4.shape[0]), dtype=f32)
    for g in range(G):
        I = _group_int_dots(qa4, qb4, g)
        s = (sA[:, g].astype(f32)[:, None] * sB[g].astype(f32)[None, :]).astype(f16).astype(f32)
        c = (c + (I * s).astype(f32)).astype(f32)
    I8 = qa8.astype(f32) @ qb8.astype(f32).T
    s = (sA8.astype(f32)[:, None] * sB8.astype(f32)[None, :]).astype(f16).astype(f32)
    c = (c + (I8 * s).astype(f32)).astype(f32)
    return c.astype(f16)


def gemm_w4a4_contract(qa4, qb4, sA, sB, qa8, qb8, sA8, sB8, kgroups: int = 1) -> np.ndarray:
    """The C-ABI arithmetic contract (include/atom_hip.h) in numpy, independent of oracle/atom_oracle.c:
    K steps = the G int4 groups in order, then the keeper (ONE dot product over its 128 columns, one de-quantisation, as the
    reference kernel: Dense_layer_gemm_i4_o16.cuh:640-691); per step s = sA * sB -- exact in float32, both are fp16 values --,
    c = fma(idot, s, c): the reference's dequant shape (scale product first, then accu += c_frag * rs_scale,
    Dense_layer_gemm_i4_o16.cuh:413-431) with one rounding per step.  The fma goes through the x87 extended format: idot * s has
    at most 46 significant bits, exact there, and one rounding of the sum to float32 is the fused result up to double rounding,
    which needs a 2^-39 coincidence per operation (checked against the C restatement in tests).  kgroups = 1: one ordered sum
    (tile kernels).  kgroups = 2 / 4: the K-group tile kernels (gemm_w4a4_f6.hip) -- the G + 1 steps are cut into ranges
    [(G+1)k/kg, (G+1)(k+1)/kg), each summed from 0, partial sums added in order.  Returns float16 [M, N]."""
    M, K4 = qa4.shape
    G = K4 // GROUP
    wide = np.longdouble
    steps = []
    for g in range(G):
        steps.append((_group_int_dots(qa4, qb4, g), sA[:, g].astype(f32), sB[g].astype(f32)))
    I8 = (qa8.astype(np.int32) @ qb8.astype(np.int32).T).astype(f32)       # |idot8| <= 128 * 128 * 127 < 2^24: exact
    steps.append((I8, sA8.astype(f32), sB8.astype(f32)))
    T = len(steps)
    total = None
    for k in range(kgroups):
        c = np.zeros((M, qb4.shape[0]), dtype=f32)
        for I, sa, sb in steps[T * k // kgroups:T * (k + 1) // kgroups]:
            s = (sa[:, None] * sb[None, :]).astype(f32)                    # exact: 11-bit x 11-bit significands
            c = (I.astype(wide) * s.astype(wide) + c.astype(wide)).astype(f32)
        total = c if total is None else (total + c).astype(f32)
    with np.errstate(over="ignore"):
        return total.astype(f16)


def quant_o4(D32: np.ndarray, ref_extrema: bool = False):
    """The _o4 epilogue (e2e/.../GEMM/DenseLayerGEMM_i4_o4.cu:704-788): asymmetric u4 per 128-col
    output group of the FP32 accumulators.  scale = (max-min)/15, zero = -min, r = 1/scale,
    q = clamp(round_half_away((x+zero)*r), 0, 15)  (fp32; the reference masks with & 0xF instead of clamping).
    We restate the *intended* min/max: the reference's local_max_min takes abs() of BOTH extrema (:73-80 of
    that file), which is wrong for any tile with negative values; its consumer de-quantises q*scale - zero
    (kernels/include/flashinfer/quantization.cuh:59-84), i.e. expects the true minimum.  See DESIGN.md.
    ref_extrema=True restates the reference CODE instead (the opt-in ATOM_O4_REF_EXTREMA mode of the library): extrema of |x|
    (:73-80, :732), scale = (max|x| - min|x|)/15, zero = -min|x| (:749-751), code = (int8)round((x+zero)*r) & 0xF with no clamp
    (:766-771; the float -> int8 cast is taken as saturating).
    Returns (u8 packed [M,N/2], half2 (scale,zero) [M, N/128, 2])."""
    D32 = np.asarray(D32, dtype=f32)
    M, N = D32.shape
    g = D32.reshape(M, N // GROUP, GROUP)
    e = np.abs(g) if ref_extrema else g
    mx = e.max(axis=-1)
    mn = e.min(axis=-1)
    scale = ((mx - mn) / f32(15)).astype(f32)
    zero = (-mn).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (f32(1.0) / scale).astype(f32)
        q = _round_half_away(((g + zero[..., None]).astype(f32) * r[..., None]).astype(f32))
    if ref_extrema:
        q = np.where(np.isnan(q), f32(-128), q)
        q = (np.clip(q, -128, 127).astype(np.int16) & 0xF).reshape(M, N)
    else:
        q = np.where(scale[..., None] == 0, f32(0), q)
        q = np.clip(q, 0, 15).astype(np.int16).reshape(M, N)
    packed = ((q[:, 0::2] & 0xF) | ((q[:, 1::2] & 0xF) << 4)).astype(np.uint8)
    sz = np.stack([scale.astype(f16), zero.astype(f16)], axis=-1)
    return packed, sz


# --------------------------------------------------------------------------- simulated path (the north-star oracle)
def sim_linear(xq16: np.ndarray, wq16: np.ndarray) -> np.ndarray:
    """QLinearLayer.forward (qLinearLayer.py:32-35): F.linear on fake-quant FP16 operands,
    FP32 accumulate, FP16 out."""
    return (xq16.astype(f32) @ wq16.astype(f32).T).astype(f16)


def sim_linear_torch(xq, wq):
    """The literal reference call, torch CPU kernels: used as bench.py's cpu_baseline ("port")."""
    import torch
    return torch.nn.functional.linear(xq, wq, None)


def kv_fake_quant_sim(x16: np.ndarray, n_bits: int = 4, clip: float = 1.0) -> np.ndarray:
    """quantize_attn_k_wrapper / quantize_attn_v_wrapper (model/quant.py:233-257): quantize_tensor(sym=False) per 128-d
    head vector, every step in FP16 opmath (one FP32 op, rounded to half) as torch does on half tensors
    (model/quant.py:143-145 max/min, :173-181 affine mapping).  x16 [..., 128] float16 -> same shape float16."""
    x = np.asarray(x16, dtype=f16)
    shape = x.shape
    w = x.reshape(-1, shape[-1]).astype(np.float32)
    h = lambda a: a.astype(f16).astype(np.float32)          # round an FP32 result to half
    qmax = np.float32((1 << n_bits) - 1)
    w_max = w.max(axis=1, keepdims=True)
    w_min = w.min(axis=1, keepdims=True)
    if clip < 1.0:
        w_max = h(w_max * np.float32(clip))
        w_min = h(w_min * np.float32(clip))
    rng = np.maximum(h(w_max - w_min), np.float32(f16(1e-5)))
    scales = h(rng / qmax)
    with np.errstate(over="ignore", invalid="ignore"):
        base = np.clip(_rne(h(-w_min / scales)), 0, qmax)
        q = np.clip(h(_rne(h(w / scales)) + base), 0, qmax)
        out = h(h(q - base) * scales)
    return out.astype(f16).reshape(shape)


# --------------------------------------------------------------------------- INT4 paged KV cache
# The reference holds no golden VECTOR for these (tests/test_batch_decode_int4.py only checks that the CUDA kernel runs) and its
# kernels cannot run here, but it ships CPU implementations of both operations as its own test oracle
# (kernels/src/flashinfer/cpu_reference.h: append_paged_kv_cache, single_quantize_mha).  PINNED against those, compiled from
# /root/reference into oracle/_ref: tests/test_oracle_ref.py (append byte for byte, decode to FP32 accuracy).
def kv_seq_len(indptr, last_page_offset, page_size, b):
    """page.cuh:170-172 / decode.cuh:510-512."""
    return (int(indptr[b + 1]) - int(indptr[b]) - 1) * page_size + int(last_page_offset[b])


def kv_append_i4(data, param, indptr, indices, last_page_offset, k, v, k_param, v_param, layer, append_indptr=None):
    """AppendPagedKVCachePrefillKernel / ...DecodeKernel (kernels/include/flashinfer/page.cuh:119-227), in place.
    data  u8  [pages, L, 2, N, P, D/2]   param f16 [pages, L, 2, N, P, 2]   (punica/utils/kvcache.py:17-26)
    k, v  u8  [T, N, D/2]                k_param, v_param f16 [T, N, 2] = (scale, zero)
    append_indptr int32 [B+1] (prefill: the LAST append_len tokens of every sequence) or None (decode: one token each)."""
    B = len(last_page_offset)
    P = data.shape[4]
    for b in range(B):
        seq_len = kv_seq_len(indptr, last_page_offset, P, b)
        if append_indptr is None:
            t0, n = b, 1
        else:
            t0, n = int(append_indptr[b]), int(append_indptr[b + 1]) - int(append_indptr[b])
        for j in range(n):
            pos = seq_len - n + j
            page = int(indices[int(indptr[b]) + pos // P])
            e = pos % P
            data[page, layer, 0, :, e, :] = k[t0 + j]
            data[page, layer, 1, :, e, :] = v[t0 + j]
            param[page, layer, 0, :, e, :] = k_param[t0 + j]
            param[page, layer, 1, :, e, :] = v_param[t0 + j]


def _dequant_u4_rows(packed, prm):
    """quantization.cuh:59-84: float(nibble) * scale - zero, element 2j in the low nibble.  packed u8 [S, D/2],
    prm f16 [S, 2] -> float64 [S, D]."""
    u = np.empty((packed.shape[0], packed.shape[1] * 2), dtype=np.float64)
    u[:, 0::2] = packed & 0xF
    u[:, 1::2] = packed >> 4
    p = prm.astype(np.float64)
    return u * p[:, 0:1] - p[:, 1:2]


def _rope_llama(x, pos, theta=1e4):
    """decode.cuh:40-72 (apply_llama_rope): pairs (i, i + D/2), freq_i = theta^(-2 (i mod D/2) / D).  x [.., S, D],
    pos [S]."""
    D = x.shape[-1]
    i = np.arange(D) % (D // 2)
    freq = (1.0 / theta) ** (2.0 * i / D)
    ang = np.asarray(pos, dtype=np.float64)[:, None] * freq[None, :]
    rot = np.concatenate([-x[..., D // 2:], x[..., :D // 2]], axis=-1)
    return x * np.cos(ang) + rot * np.sin(ang)


def batch_decode_i4(q16, data, param, indptr, indices, last_page_offset, layer, theta=1e4):
    """BatchDecodeWithPagedKVCacheKernel (decode.cuh:480-676) as called by flashinfer_impl.cuh:9-46: RoPE (llama) on q at
    position seq_len-1 and on every de-quantised key at its own position, sm_scale = 1/sqrt(D), softmax, P.V on the
    de-quantised values; FP64 here.  q16 f16 [B, N, D] -> float64 [B, N, D]."""
    B, N, D = q16.shape
    P = data.shape[4]
    out = np.zeros((B, N, D), dtype=np.float64)
    for b in range(B):
        S = kv_seq_len(indptr, last_page_offset, P, b)
        pages = [int(x) for x in indices[int(indptr[b]):int(indptr[b + 1])]]
        for h in range(N):
            kp = np.concatenate([data[pg, layer, 0, h] for pg in pages], axis=0)[:S]
            vp = np.concatenate([data[pg, layer, 1, h] for pg in pages], axis=0)[:S]
            kq = np.concatenate([param[pg, layer, 0, h] for pg in pages], axis=0)[:S]
            vq = np.concatenate([param[pg, layer, 1, h] for pg in pages], axis=0)[:S]
            kf = _rope_llama(_dequant_u4_rows(kp, kq), np.arange(S), theta)
            vf = _dequant_u4_rows(vp, vq)
            qf = _rope_llama(q16[b, h].astype(np.float64)[None, :], [S - 1], theta)[0]
            s = kf @ qf / np.sqrt(D)
            p = np.exp(s - s.max())
            out[b, h] = (p / p.sum()) @ vf
    return out
