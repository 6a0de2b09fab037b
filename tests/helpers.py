"""Shared helpers for the GPU parity tests (checker side only)."""
import numpy as np
import torch

from oracle import atom_oracle as O


def bits16(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float16)).view(np.uint16)


def t2n(t):
    return t.detach().cpu().numpy()


def scales_plain(t, M, layout):
    """Device scale tensor ([G, ld] or [ld]) -> plain numpy [G, M] / [M]."""
    a = t2n(t)
    if layout == "ref":
        return O.scales_from_ref_layout(a, M)
    return a[..., :M]


def rand_act(M, H, seed, outliers=True):
    g = np.random.default_rng(seed)
    x = g.standard_normal((M, H)).astype(np.float32)
    if outliers:
        cols = g.permutation(H)[:128]
        x[:, cols] *= 20.0
    return x.astype(np.float16)


def rand_gemm_operands(M, N, K, seed, device="cuda"):
    """Synthetic operands per SURVEY 8(d): uniform int4/int8 codes, scales ~U(0.005,0.05), weight scales equal
    for channel pairs.  Returns dict of numpy arrays (codes unpacked) -- pack/move with to_device()."""
    g = np.random.default_rng(seed)
    K4 = K - 128
    G = K4 // 128
    d = dict(
        qa4=g.integers(-8, 8, size=(M, K4), dtype=np.int8),
        qb4=g.integers(-8, 8, size=(N, K4), dtype=np.int8),
        qa8=g.integers(-128, 128, size=(M, 128), dtype=np.int16).astype(np.int8),
        qb8=g.integers(-128, 128, size=(N, 128), dtype=np.int16).astype(np.int8),
        sA=g.uniform(0.005, 0.05, size=(M, G)).astype(np.float16),
        sA8=g.uniform(0.005, 0.05, size=(M,)).astype(np.float16),
        sB8=g.uniform(0.005, 0.05, size=(N,)).astype(np.float16),
    )
    sb = g.uniform(0.005, 0.05, size=(G, N // 2)).astype(np.float16)
    d["sB"] = np.repeat(sb, 2, axis=1)
    return d


def to_device(d, layout, device="cuda"):
    """numpy operand dict -> the 8 device tensors dense_layer_gemm_i4_fp16 takes."""
    M = d["qa4"].shape[0]
    sA = np.ascontiguousarray(d["sA"].T)              # [G, M]
    if layout == "ref":
        sA = O.scales_to_ref_layout(sA)
        sA8 = O.scales_to_ref_layout(d["sA8"])
    else:
        sA8 = d["sA8"]
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return (f(O.pack_int4(d["qa4"])), f(O.pack_int4(d["qb4"])), f(sA), f(d["sB"]), f(d["qa8"]), f(d["qb8"]),
            f(sA8), f(d["sB8"]))


def gemm_ref_torch_f64(d, device="cuda"):
    """Independent torch FP64 evaluation of the GEMM definition on the GPU (checker for sizes the numpy oracle
    would take minutes on)."""
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    qa4, qb4 = f(d["qa4"]).double(), f(d["qb4"]).double()
    sA, sB = f(d["sA"]).double(), f(d["sB"]).double()
    M, K4 = qa4.shape
    G = K4 // 128
    out = torch.zeros((M, qb4.shape[0]), dtype=torch.float64, device=device)
    for g in range(G):
        sl = slice(g * 128, (g + 1) * 128)
        out += (qa4[:, sl] @ qb4[:, sl].T) * sA[:, g:g + 1] * sB[g:g + 1, :]
    out += (f(d["qa8"]).double() @ f(d["qb8"]).double().T) * f(d["sA8"]).double()[:, None] * f(d["sB8"]).double()[None, :]
    return out


def assert_gemm_close(d_hip, d_exact, what=""):
    """fp16 output rounding is 2^-11 relative; FP32 accumulation noise is far below that."""
    d_hip = np.asarray(d_hip, dtype=np.float64)
    d_exact = np.asarray(d_exact, dtype=np.float64)
    rms = np.sqrt((d_exact ** 2).mean())
    err = np.abs(d_hip - d_exact)
    tol = 1e-3 * np.abs(d_exact) + 1e-3 * rms
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} elements off, max err {err.max():.4g}, rms {rms:.4g}"
    rel = np.linalg.norm(d_hip - d_exact) / max(np.linalg.norm(d_exact), 1e-30)
    assert rel < 5e-4, f"{what}: relative Frobenius error {rel:.3g}"


def wide_codes(q4):
    """int8 codes [M, K4] -> the native wide activation format (include/atom_hip.h): byte[m, 32c+16p+j] = 16*q[m, 32c+2j+p]."""
    q = np.asarray(q4, dtype=np.int8)
    M, K4 = q.shape
    blk = q.reshape(M, K4 // 32, 16, 2)                      # [.., j, p]
    out = np.transpose(blk, (0, 1, 3, 2)).reshape(M, K4)     # [.., p, j]
    return (out.astype(np.int16) * 16).astype(np.int8)


_BF6_MAG = [0x00, 0x0C, 0x10, 0x12, 0x14, 0x15, 0x16, 0x17, 0x18]


def f6_codes(q4, scales=None):
    """int8 codes [M, K4] (+ fp16 scales [M, G]) -> the F6 operand buffer uint8 [G][round_up(M,256)][104]
    (include/atom_hip.h, ATOM_AB_F6): 6-bit BF6 (E3M2) fields, little-endian, scale at byte 96 (fp16) and 100 (fp32); pad rows zero."""
    q = np.asarray(q4, dtype=np.int64)
    M, K4 = q.shape
    G = K4 // 128
    rp = (M + 255) // 256 * 256
    lut = np.array([(0x20 if v < 0 else 0) | _BF6_MAG[abs(v)] for v in range(-8, 8)], dtype=np.uint64)
    c = lut[q + 8].reshape(M, G, 32, 4)                                   # 4 fields = 24 bits = 3 bytes
    w = c[..., 0] | (c[..., 1] << 6) | (c[..., 2] << 12) | (c[..., 3] << 18)
    b = np.stack([(w >> (8 * i)) & 0xFF for i in range(3)], axis=-1).astype(np.uint8).reshape(M, G, 96)
    out = np.zeros((G, rp, 104), dtype=np.uint8)
    out[:, :M, :96] = np.transpose(b, (1, 0, 2))
    if scales is not None:
        sc16 = np.ascontiguousarray(np.asarray(scales, dtype=np.float16).T)
        out[:, :M, 96:98] = sc16.view(np.uint8).reshape(G, M, 2)
        out[:, :M, 100:104] = sc16.astype(np.float32).view(np.uint8).reshape(G, M, 4)      # the same scale as fp32
    return out


def f6_fields(buf):
    """F6 buffer [G, rows, 104] -> the 6-bit fields [G, rows, 128] with the code of -0 (0x20) folded onto +0."""
    b = np.asarray(buf, dtype=np.uint64)[..., :96].reshape(*buf.shape[:2], 32, 3)
    w = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
    f = np.stack([(w >> (6 * i)) & 0x3F for i in range(4)], axis=-1).reshape(*buf.shape[:2], 128).astype(np.uint8)
    f[f == 0x20] = 0
    return f


def recover_fake_quant(v, qmax, group):
    """Integer codes and the per-(row, group) fp16 scale of a fake-quantised tensor v = half(code * s) (what the reference's
    quantisers return, model/quant.py:141-181): the largest-code candidate s = half(max|v| / k), k = qmax + 1 .. 1 (and its fp16
    neighbours), for which every element is reproduced EXACTLY by an integer code in [-qmax - 1, qmax].  Where several (code, s)
    pairs describe the same numbers they differ by a power of two and the GEMM contract gives identical sums.  Returns
    (codes int32 [rows, cols], scales fp16 [rows, groups]); raises if a group is on no such grid.  (The same search as
    tests/golden/gen_golden_block7b.py:recover_scales, which wrote the wide 7B fixture with it.)"""
    v = np.asarray(v, dtype=np.float16)
    r, h = v.shape
    x = v.astype(np.float32).reshape(r, h // group, group)
    vmax = np.abs(x).max(-1)
    best = np.zeros(vmax.shape, dtype=np.float16)
    done = vmax == 0
    best[done] = 1.0
    for k in range(qmax + 1, 0, -1):
        base = (vmax / k).astype(np.float16)
        for dlt in (0, 1, -1):
            cand = (base.view(np.int16) + dlt).view(np.float16)
            cf = cand.astype(np.float32)
            with np.errstate(divide="ignore", invalid="ignore"):
                c = np.rint(x / cf[..., None])
            ok = (np.abs(c).max(-1) <= qmax + 1) & (c.max(-1) <= qmax) & np.isfinite(cf) & (cf > 0)
            ok &= ((c * cf[..., None]).astype(np.float16) == x.astype(np.float16)).all(-1)
            take = ok & ~done
            best[take] = cand[take]
            done |= take
    assert done.all(), "fake-quantised group not reproducible by integer codes: %d of %d" % ((~done).sum(), done.size)
    codes = np.rint(x / best.astype(np.float32)[..., None]).astype(np.int32).reshape(r, h)
    return codes, best
