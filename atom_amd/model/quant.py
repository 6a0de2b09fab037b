"""HIP-backed mirror of the reference's ``model/quant.py`` (same public names and signatures).

Reference semantics (file:line relative to /root/reference/model/quant.py):
  quantize_tensor                 :118-183   uniform affine fake-quant of [groups, group_size] rows
  quantize_tensor_channel_group   :68-107    weight wrapper: ``channel_group`` adjacent rows share a scale
  quantize_activation_wrapper     :187-231   dynamic per-token activation quant with INT8 keeper columns
  quantize_attn_{k,v}_wrapper     :233-257   asymmetric 4-bit per 128-d head vector (KV cache)
  Quantizer                       :259-305

Hot path: with the paper configuration (abits=4, a_sym, act_group_size=128, keeper=128, keeper_precision=3, int) and
a CUDA fp16 input, ``quantize_activation_wrapper`` runs ONE fused HIP kernel (atom_reorder_quant_f16, sim arithmetic)
that emits packed INT4/INT8 codes + scales AND the de-quantised fp16 tensor the reference returns; the codes ride along
on the returned tensor (``._atom_codes``) so that ``QLinearLayer.forward`` can feed the W4A4 GEMM without re-quantising
(re-quantising a clipped fake-quant tensor is not idempotent).  There is no CPU fallback for that configuration.
Other configurations use plain torch ops on the tensor's own device, exactly following the reference formulas.
"""
from __future__ import annotations

from functools import partial  # noqa: F401  (re-exported: callers do ``partial(quantize_activation_wrapper, args=args)``)

import torch
from torch import nn

from .. import ops as _ops
from .._lib import AtomHipError

GROUP = 128


class ActCodes:
    """Integer-domain view of a fake-quantised activation: what the W4A4 GEMM consumes."""
    __slots__ = ("o8", "o4", "s8", "s4", "rows", "hidden", "layout", "wide")

    def __init__(self, o8, o4, s8, s4, rows, hidden, layout="plain", wide=False):
        self.o8, self.o4, self.s8, self.s4 = o8, o4, s8, s4
        self.rows, self.hidden, self.layout = rows, hidden, layout
        self.wide = wide          # o4 is the native int8 (code*16, de-interleaved) format instead of packed nibbles


def want_wide_codes(rows: int):
    """Which activation-code format the fused quantisers should emit for a batch of ``rows`` tokens:
    "f6"   rows > 128: the BF6 group-major format of the block-scaled-MFMA kernels (tile geometries picked by shape; measured
           against the INT8 kernels over Llama projection shapes in profiles/r01_f6_dispatch.txt: ahead from 256 rows up, 1.3-1.4x at
           1k-2k rows; round 5: the mid-size-batch kernel runs 129 .. 256 rows in ~10 us at 4096 x 4096 where the decode-batch kernel
           takes 9.4 (128 rows) .. 15.5 (256), profiles/r05/mid_f6c.txt);
    False  rows <= 128: packed nibbles -- the decode kernels (gemv / decode-batch GEMM) take only these.
    (True = pre-widened int8 codes for the INT8 tile kernels is still accepted by every op; nothing here asks for it.)"""
    if rows > 128:
        return "f6"
    return False


MAX_HOT_HIDDEN = 16384     # the fused activation kernels take hidden sizes up to this (quant_kernels.hip launch_act_quant)


def is_hot_act_config(args, hidden: int) -> bool:
    """The configuration the HIP activation kernels implement (scripts/run_atom_ppl.sh:11-15).  Wider layers (the MLP
    intermediate sizes 17920 / 22016 / 28672 of Llama-30B / 65B / 70B) take the reference's algorithm in torch ops instead."""
    return (args.abits == 4 and bool(args.a_sym) and args.act_group_size == GROUP and args.keeper == GROUP
            and getattr(args, "keeper_precision", 0) == 3 and getattr(args, "quant_type", "int") == "int"
            and not getattr(args, "exponential", False) and hidden % GROUP == 0 and 2 * GROUP <= hidden <= MAX_HOT_HIDDEN)


def attach_codes(t: torch.Tensor, codes: ActCodes) -> torch.Tensor:
    t._atom_codes = codes
    return t


def get_codes(t: torch.Tensor):
    return getattr(t, "_atom_codes", None)


# ------------------------------------------------------------------------------------------------ FP8 keepers
def fake_quantize_quarter_E5M2(w: torch.Tensor) -> torch.Tensor:
    """FP16 -> E5M2 -> FP16 (reference :9-24 does the same cast with hand-rolled bit twiddling; keeper_precision=1,
    outside the INT8-keeper hot path)."""
    assert w.dtype == torch.float16
    return w.to(torch.float8_e5m2).to(torch.float16)


def fake_quantize_quarter_E4M3(w: torch.Tensor) -> torch.Tensor:
    """FP16 -> E4M3 -> FP16 (reference :28-66; note the reference saturates at 480, OCP e4m3fn at 448)."""
    assert w.dtype == torch.float16
    return w.clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float16)


# ------------------------------------------------------------------------------------------------ generic fake quant
@torch.no_grad()
def quantize_tensor(w: torch.Tensor, n_bits, group_size, tiling, sym, clip_ratio=1.0, exponential=False,
                    quant_type="int") -> torch.Tensor:
    """Uniform fake quantisation of rows (reference :118-183), all arithmetic in w's dtype like the reference."""
    if tiling > 0:
        raise AssertionError("16x16 block-wise quantization is abandoned in the reference")
    assert n_bits < 16
    assert quant_type in ("int", "fp"), "Options should be in [int, fp]"
    shape = w.shape
    w = w.squeeze()
    if group_size > 0:
        assert w.shape[-1] % group_size == 0
        w = w.reshape(-1, group_size)
    assert w.dim() == 2, "expected [num_groups, group_size]"
    if quant_type == "fp":
        return _fake_quant_fp4(w, n_bits).reshape(shape)
    if exponential:
        return _fake_quant_exponent(w, n_bits, sym).reshape(shape)
    if sym:
        hi, lo = 2 ** (n_bits - 1) - 1, -(2 ** (n_bits - 1))
        amax = w.abs().amax(dim=-1, keepdim=True).clamp(min=1e-5)
        if clip_ratio < 1.0:
            amax = amax * clip_ratio
        scale = amax / hi
        out = torch.clamp(torch.round(w / scale), lo, hi) * scale
    else:
        hi, lo = 2 ** n_bits - 1, 0
        vmax, vmin = w.amax(dim=-1, keepdim=True), w.amin(dim=-1, keepdim=True)
        if clip_ratio < 1.0:
            vmax, vmin = vmax * clip_ratio, vmin * clip_ratio
        scale = (vmax - vmin).clamp(min=1e-5) / hi
        zero = torch.round(-vmin / scale).clamp_(min=lo, max=hi)
        out = (torch.clamp(torch.round(w / scale) + zero, lo, hi) - zero) * scale
    return out.reshape(shape)


_FP4_LEVELS = (0.0, 0.0625, 2.0, 3.0, 4.0, 6.0, 8.0, 12.0)     # bitsandbytes "fp4" magnitudes (x / 12 after absmax scaling)


def _fake_quant_fp4(w: torch.Tensor, n_bits) -> torch.Tensor:
    """quant_type="fp" (reference :134-138: bitsandbytes quantize_fp4 / dequantize_fp4 with blocksize = the group): every row is
    scaled by its absmax and each value replaced by the nearest of the 16 FP4 levels {+-0, +-1/192, +-1/6, +-1/4, +-1/3, +-1/2,
    +-2/3, +-1} x absmax.  Not part of Atom's configuration (the hot path is INT4); bitsandbytes is not installed here, so this
    branch is a restatement of its published data type, unpinned (ties between two levels may round differently)."""
    assert n_bits == 4, "Only support FP4 quantization. You can add more by using bnb library."
    f = w.float()
    amax = f.abs().amax(dim=-1, keepdim=True)
    x = torch.where(amax > 0, f / amax, torch.zeros_like(f))
    lv = torch.tensor(_FP4_LEVELS, dtype=torch.float32, device=w.device) / 12.0
    idx = (x.abs().unsqueeze(-1) - lv).abs().argmin(dim=-1)
    return (torch.sign(x) * lv[idx] * amax).to(w.dtype)


def _fake_quant_exponent(w: torch.Tensor, n_bits, sym) -> torch.Tensor:
    """exponential=True (reference :146-164): an exponent-only format -- magnitudes become scales * 2^e, e in [0, 2^(n_bits-1) - 1],
    with e = floor(log2(|w| / scales)) rounded up where the mantissa exceeds 1.5; asymmetric: around the mid-point of the range.
    "not used in Atom" (reference :115); arithmetic in w's dtype like the reference."""
    q_max = 2 ** (2 ** (n_bits - 1) - 1)
    if sym:
        scales = w.abs().amax(dim=-1, keepdim=True).clamp(min=1e-5)
        base = torch.zeros_like(scales)
    else:
        hi, lo = w.amax(dim=-1, keepdim=True), w.amin(dim=-1, keepdim=True)
        scales = (hi - lo) * torch.tensor(0.5)
        base = (hi + lo) * torch.tensor(0.5)
    scales = scales / q_max
    c = w - base
    sign = torch.sign(c)
    lg = torch.log2((torch.abs(c) / scales).clamp(min=1, max=q_max))
    e = torch.floor(lg)
    e = e + (lg - e > torch.log2(torch.tensor(1.5))).int()
    return (2 ** e) * sign * scales + base


@torch.no_grad()
def quantize_tensor_channel_group(W: torch.Tensor, n_bits, group_size, tiling, sym, channel_group=1, clip_ratio=1.0,
                                  exponential=False, quant_type="int") -> torch.Tensor:
    """Weight fake-quant, ``channel_group`` adjacent output rows sharing one scale per column slice (reference :68-107).
    (The packed/HIP weight path is QLinearLayer.quant; this torch version serves non-hot configurations.)"""
    assert W.is_contiguous() and n_bits < 16
    if group_size == 0:
        return quantize_tensor(W, n_bits, 0, tiling, sym, exponential=exponential)
    assert W.shape[-1] % group_size == 0
    for c0 in range(0, W.shape[1], group_size):
        blk = W[:, c0:c0 + group_size]
        if channel_group > 1:
            blk = blk.reshape(W.shape[0] // channel_group, -1).contiguous()
        blk = quantize_tensor(blk, n_bits, 0, tiling, sym, clip_ratio, exponential, quant_type)
        W[:, c0:c0 + group_size] = blk.reshape(-1, group_size)
    return W.contiguous()


# ------------------------------------------------------------------------------------------------ activations
def _reorder_index_i16(index: torch.Tensor | None, device):
    if index is None:
        return None
    cached = getattr(index, "_atom_i16", None)
    if cached is None or cached.device != device:
        cached = index.to(device=device, dtype=torch.int16)
        try:
            index._atom_i16 = cached
        except Exception:
            pass
    return cached


@torch.no_grad()
def hip_act_quant(x: torch.Tensor, args, reorder_index: torch.Tensor | None = None) -> torch.Tensor:
    """[.., K] fp16 CUDA -> fake-quant fp16 tensor carrying ActCodes; optional fused channel gather."""
    if not x.is_cuda:
        raise AtomHipError("W4A4 activation quantisation needs a GPU tensor: the Atom hot path has no CPU fallback")
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    wide = want_wide_codes(x2.shape[0])
    o8, o4, s8, s4, xq = _ops.reorder_fp16_i4(x2, _reorder_index_i16(reorder_index, x.device), quant_mode="sim",
                                              clip=float(args.a_clip_ratio), scale_layout="plain",
                                              return_dequant=True, wide_codes=wide)
    return attach_codes(xq.view(shape), ActCodes(o8, o4, s8, s4, x2.shape[0], shape[-1], wide=wide))


@torch.no_grad()
def quantize_activation_wrapper(x: torch.Tensor, args) -> torch.Tensor:
    """Dynamic per-token activation quantisation with mixed-precision keeper columns (reference :187-231).
    Unlike the reference this does not zero the keeper columns of the CALLER's tensor as a side effect."""
    if args.abits >= 16:
        return x
    hidden = x.shape[-1]
    if x.dtype == torch.float16 and is_hot_act_config(args, hidden):
        return hip_act_quant(x, args)
    # non-hot configurations: the reference algorithm in torch ops on x's device
    shape = x.shape
    x = x.reshape(-1, hidden).clone()
    assert args.act_group_size == 0 or hidden % args.act_group_size == 0
    keep = None
    if args.keeper > 0:
        keep = x[:, -args.keeper:].clone().contiguous()
        kp = getattr(args, "keeper_precision", 0)
        if kp == 1:
            keep = fake_quantize_quarter_E5M2(keep)
        elif kp == 2:
            keep = fake_quantize_quarter_E4M3(keep)
        elif kp == 3:
            keep = quantize_tensor(keep, n_bits=8, group_size=0, tiling=0, sym=True)
        x[:, -args.keeper:] = 0
    x = quantize_tensor(x, n_bits=args.abits, group_size=args.act_group_size, tiling=args.tiling, sym=args.a_sym,
                        clip_ratio=args.a_clip_ratio, quant_type=getattr(args, "quant_type", "int"))
    if keep is not None:
        x[:, -args.keeper:] = keep
    return x.view(shape)


@torch.no_grad()
def _quantize_head_vectors(w: torch.Tensor, args) -> torch.Tensor:
    assert w.shape[-1] == 128, "KV cache quantization is per 128-d head vector"
    if w.is_cuda and w.dtype == torch.float16 and w.dim() == 4 and w.stride(-1) == 1 and 2 <= args.abits <= 8 \
            and all(st % 8 == 0 for st in w.stride()[:3]):
        return _ops.kv_fake_quant(w, int(args.abits), float(args.kv_clip_ratio))      # one fused HIP kernel
    shape = w.shape
    out = quantize_tensor(w.reshape(-1, 128), n_bits=args.abits, group_size=0, tiling=0, sym=False,
                          clip_ratio=args.kv_clip_ratio)
    return out.view(shape)


def quantize_attn_v_wrapper(w: torch.Tensor, args) -> torch.Tensor:
    """[bsz, heads, seq, 128] -> asymmetric abits fake quant per head vector (reference :233-244)."""
    return _quantize_head_vectors(w, args)


def quantize_attn_k_wrapper(w: torch.Tensor, args) -> torch.Tensor:
    """Reference :246-257 (applied BEFORE RoPE, qLlamaLayer.py:248-249)."""
    return _quantize_head_vectors(w, args)


class Quantizer(nn.Module):
    """Holder of the activation-quant callable (reference :259-305).  Atom is dynamic: ``forward(x) = act_quant(x)``;
    the static branch of the reference is dead code there and is not implemented."""

    def __init__(self, args) -> None:
        super().__init__()
        self.register_buffer("scales", None)
        self.args = args
        self.act_quant = lambda x: x          # configured from outside (modelutils_llama.py:96-120)

    @torch.no_grad()
    def forward(self, hidden_states):
        if self.args.static is False or self.scales is None:
            return self.act_quant(hidden_states)
        # static scales (reference :274-290; "Atom is dynamic quantization", so only non-Atom configurations get here): the columns
        # behind the first `keeper` ones are rounded onto pre-computed per-group scales, in place like the reference
        a = self.args
        shape = hidden_states.shape
        assert a.a_sym is True, "Only support statically symmetric quantization"
        assert a.act_group_size == 0 or (shape[-1] - a.keeper) % a.act_group_size == 0
        hs = hidden_states.view(-1, shape[-1])
        sel = hs[:, a.keeper:].clone()
        if a.act_group_size > 0:
            sel = sel.reshape(-1, a.act_group_size)
        assert self.scales.numel() == sel.shape[-2], "Scales and selected states must have the same dimension"
        sel = torch.clamp(torch.round(sel / self.scales), self.q_min, self.q_max) * self.scales
        hs[:, a.keeper:] = sel.reshape(-1, shape[-1] - a.keeper)
        return hs.view(shape)

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        if self.scales is not None:
            self.scales = self.scales.to(*args, **kwargs)
        return self

    def configure(self, func, scales):
        if self.args.static is False:
            self.act_quant = func
            return
        assert scales is not None, "Scales is None"
        self.register_buffer("scales", scales)
        self.q_min = -(2 ** (self.args.abits - 1))
        self.q_max = 2 ** (self.args.abits - 1) - 1

    # --- helpers for the fused layers in qLlamaLayer.py ------------------------------------------------------
    def hot_args(self, hidden: int):
        """args if this quantizer is configured with quantize_activation_wrapper in the HIP configuration."""
        f = self.act_quant
        if getattr(f, "func", None) is quantize_activation_wrapper:
            a = f.keywords.get("args") if f.keywords else None
            if a is not None and a.abits < 16 and is_hot_act_config(a, hidden):
                return a
        return None
