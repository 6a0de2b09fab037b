"""HIP-backed mirror of the reference's ``model/qLinearLayer.py`` (same class / function names and signatures).

``QLinearLayer`` keeps the reference contract -- a mutable fp16 ``weight`` buffer [N, K] that ``reorder()`` permutes,
``quant()`` fake-quantises in place and GPTQ may overwrite (``layer.weight.data = Q``, gptq.py:331) -- and ADDS the
packed INT4/INT8 form the W4A4 GEMM consumes (the bridge the reference lacks):

    quant()    ->  atom_quant_weight_w4: one HIP kernel writes B4 / B8 / sB / sB8 and the fake-quant fp16 weight
    forward(x) ->  atom_gemm_w4a4_f16 when x carries activation codes (quant.ActCodes) and the packed weight is
                   current; ``F.linear`` (the reference's own forward, qLinearLayer.py:32-35) otherwise, e.g. for
                   16-bit configurations or a weight that is not (yet) quantised.
    GPTQ       ->  atom_pack_weight_w4 on the first forward after ``layer.weight.data = Q`` (codes/scales recovered).
"""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn

from .. import ops as _ops
from .quant import (fake_quantize_quarter_E4M3, fake_quantize_quarter_E5M2, get_codes,  # noqa: F401
                    quantize_tensor, quantize_tensor_channel_group)

GROUP = 128


def find_qlinear_layers(module, name=''):
    """reference qLinearLayer.py:5-14 (exact-type match, only layers with enable_quant)."""
    if type(module) == QLinearLayer:
        return {name: module} if module.enable_quant else {}
    found = {}
    for child_name, child in module.named_children():
        found.update(find_qlinear_layers(child, name=f"{name}.{child_name}" if name else child_name))
    return found


def _is_hot_weight_config(args, n, k) -> bool:
    return (args.wbits == 4 and bool(args.w_sym) and args.weight_group_size == GROUP and args.keeper == GROUP
            and getattr(args, "keeper_precision", 0) == 3 and args.weight_channel_group in (1, 2)
            and getattr(args, "quant_type", "int") == "int" and not getattr(args, "exponential", False)
            and k % GROUP == 0 and k >= 2 * GROUP and n % 64 == 0)


class QLinearLayer(nn.Module):
    # Weight memory beside the reference's fp16 fake-quant `weight` (2 B per weight, part of its contract: GPTQ and the evaluation
    # flow read and rewrite it): packed nibbles 0.5 B (+ scales ~0.02) for the decode kernels, and from the first batch of >= 256
    # rows the F6 form 0.8125 B + 0.03 B of fp32 scales for the prefill kernels.  keep_packed_with_f6 = False releases the packed
    # INT4 codes once the F6 form exists (a layer that only sees prefill batches then holds 6.75 bit per weight instead of 10.9);
    # a later decode-size batch re-packs them from `weight` (one atom_pack_weight_w4 launch).
    keep_packed_with_f6 = True
    # keep_f6 = False (round 6): the weight stays at 4 bits in HBM on the prefill path too -- the reference's weight_int4 layout, 0.52 B
    # per weight incl. scales (punica/models/llama.py:35-59) -- and a prefill batch re-codes it to the BF6 form into a TRANSIENT buffer
    # (one bandwidth-bound launch, freed after the GEMM) instead of keeping 0.84 B per weight beside it.  Same kernels, same bits;
    # the price is the re-coding per call (and, in the MLP, re-building the interleaved gate / up operand): the seven GEMMs of a
    # Llama-7B block take +3.4 % at the 65,536 tokens of BASELINE config 4 (the block +0.7 %), +14 % at 16,384 tokens, +60 % at 4,096
    # (profiles/r06/ab_nibble_weights.txt).  Default True (speed at every batch size); False for deployments that count the bytes.
    keep_f6 = True
    # Forwards that arrived with INT4 activation codes and a hot-path configuration but were served by F.linear because the weight is
    # not on the INT4-g128 / INT8 grid (pack_weight_w4 found off-grid blocks): the reference's own arithmetic, but not the HIP path.
    # Expected while GPTQ collects Hessians on still-unquantised weights; anywhere else it means the layer was never quantised.
    # Class-wide total here, per layer in `self.offgrid_fallbacks`; the first one of every layer shape also raises a RuntimeWarning.
    offgrid_fallbacks_total = 0

    def weight_bytes(self):
        """Bytes currently held for this layer's weight, by form."""
        d = {"fp16_fake_quant": self.weight.numel() * self.weight.element_size(), "packed_int4": 0, "keeper_and_scales": 0, "f6": 0}
        if self._packed is not None:
            b4, b8, sb, sb8 = self._packed
            d["packed_int4"] = 0 if b4 is None else b4.numel() * b4.element_size()
            d["keeper_and_scales"] = sum(t.numel() * t.element_size() for t in (b8, sb, sb8))
        if self._f6 is not None:
            t = self._f6[1]
            buf = getattr(t, "atom_f6s", t)
            d["f6"] = buf.numel() * buf.element_size()
        return d

    def __init__(self, originalLayer: nn.Linear, args, enable_quant: bool = True):
        super().__init__()
        self.args = args
        self.register_buffer('weight', originalLayer.weight)
        self.enable_quant = enable_quant
        if originalLayer.bias is not None:
            self.register_buffer('bias', originalLayer.bias)
        else:
            self.bias = None
        self._packed = None          # (B4, B8, sB, sB8)
        self._packed_key = None      # identity of the fp16 weight the packed form was made from
        self._unpackable_key = None  # identity of a weight that pack_weight_w4 found to be off the grid
        self._f6 = None              # (packed B4 it was made from, BF6 repack) for the block-scaled-MFMA kernel
        self.offgrid_fallbacks = 0   # forwards of THIS layer served by F.linear because its weight is off the grid
        self._offgrid_blocks = 0     # off-grid (channel_group x 128) blocks of the weight `_unpackable_key` names

    # ------------------------------------------------------------------------------------------------ forward
    def _weight_key(self):
        w = self.weight
        return (w.data_ptr(), w._version, tuple(w.shape), w.device)

    def packed_weight(self, need_codes: bool = True):
        """Packed operands if they are current for ``self.weight``, else None.  A weight that was rewritten since the
        last packing (GPTQ: ``layer.weight.data = Q``, gptq.py:331 -- ``quant()`` is never called on that path,
        modelutils_llama.py:224-258) is packed lazily, once per weight version, by recovering its codes and scales
        (atom_pack_weight_w4); a weight that is not on the INT4-g128 / INT8 grid (e.g. still unquantised while GPTQ
        collects its Hessian) is remembered as such and served by ``F.linear``."""
        key = self._weight_key()
        if self._packed is not None and self._packed_key == key and (need_codes is False or self._packed[0] is not None):
            return self._packed
        if self._unpackable_key == key:
            return None
        a = self.args
        n, k = self.weight.shape
        if (self.enable_quant and self.weight.is_cuda and self.weight.dtype == torch.float16
                and self.weight.is_contiguous() and _is_hot_weight_config(a, n, k)):
            b4, b8, sb, sb8, bad = _ops.pack_weight_w4(self.weight, int(a.weight_channel_group), strict=False)
            if bad == 0:
                self._packed = (b4, b8, sb, sb8)
                self._packed_key = key
                return self._packed
            self._offgrid_blocks = int(bad)
        else:
            self._offgrid_blocks = 0
        self._unpackable_key = key
        return None

    def release_codes(self):
        """keep_packed_with_f6 = False: drop the packed INT4 codes once a wider form made from them exists (the F6 operand of this
        layer, or the fused gate/up operand of the MLP that owns it); the keeper and the scales stay."""
        if not self.keep_packed_with_f6 and self._packed is not None and self._packed[0] is not None:
            self._packed = (None,) + tuple(self._packed[1:])

    @torch.no_grad()
    def forward(self, x):
        codes = get_codes(x)
        packed = self.packed_weight(need_codes=codes.wide != "f6" or self._f6 is None) if codes is not None else None
        if packed is not None and codes.hidden == self.weight.shape[1] and x.is_cuda:
            b4, b8, sb, sb8 = packed
            o4, wide = codes.o4, codes.wide
            if not wide and _ops.gemm_recodes_cached(o4.shape[0], self.weight.shape[0], self.weight.shape[1]):
                # packed codes of a batch the BF6 kernels serve faster with the weight's BF6 form at hand (large N x K below 129 rows):
                # re-code the activation and use this layer's BF6 weight, instead of a second copy in the GEMM op's per-weight cache
                o4, wide = _ops.repack_act_f6(o4.view(torch.uint8), codes.s4, scale_layout=codes.layout), "f6"
            if wide == "f6" and not self.keep_f6:         # 4-bit weights only: the BF6 form is made per call and dropped
                if b4 is None:
                    b4 = self.packed_weight(need_codes=True)[0]
                self._f6 = None
                b4 = _ops.repack_weight_f6(b4, sb)
            elif wide == "f6":                            # BF6 operands: the weight is repacked once per packed form
                if self._f6 is None or self._f6[0] != self._packed_key:
                    if b4 is None:
                        b4 = self.packed_weight(need_codes=True)[0]
                    self._f6 = (self._packed_key, _ops.repack_weight_f6(b4, sb))
                    self.release_codes()                  # keep_packed_with_f6 = False (see the class comment)
                b4 = self._f6[1]
            y = _ops.dense_layer_gemm_i4_fp16(o4, b4, codes.s4, sb, codes.o8, b8, codes.s8, sb8,
                                              scale_layout=codes.layout, a_wide=wide)
            y = y.view(*x.shape[:-1], self.weight.shape[0])
            if self.bias is not None:
                y = y + self.bias
            return y
        if codes is not None and packed is None and self._offgrid_blocks and self._unpackable_key == self._weight_key():
            self.offgrid_fallbacks += 1
            QLinearLayer.offgrid_fallbacks_total += 1
            if self.offgrid_fallbacks == 1:
                n, k = self.weight.shape
                warnings.warn(f"QLinearLayer {n}x{k}: {self._offgrid_blocks} weight blocks are not on the INT4-g128 / INT8 grid -- this "
                              "layer runs F.linear on the fake-quantised tensors, not the HIP W4A4 kernels (expected only while GPTQ "
                              "collects Hessians on unquantised weights; see QLinearLayer.offgrid_fallbacks_total)", RuntimeWarning,
                              stacklevel=2)
        return torch.functional.F.linear(x, self.weight, self.bias)

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        self.weight = self.weight.to(*args, **kwargs)
        if self._packed is not None:
            dev = self.weight.device
            if dev.type == "cuda":
                # released INT4 codes (keep_packed_with_f6 = False) are None entries
                self._packed = tuple(None if t is None else t.to(dev) for t in self._packed)
                self._packed_key = self._weight_key()
            else:
                self._packed_key = None      # packed form stays on its GPU; re-keyed when the weight comes back
        if self._f6 is not None and self._f6[0] != self._packed_key:
            self._f6 = None                  # made for another device / weight: rebuilt by the next prefill batch
        return self

    # ------------------------------------------------------------------------------------------------ quant
    @torch.no_grad()
    def quant(self):
        """reference qLinearLayer.py:42-78.  Hot configuration: one HIP kernel; otherwise the reference steps in torch."""
        a = self.args
        if a.wbits >= 16:
            return
        n, k = self.weight.shape
        if self.weight.is_cuda and self.weight.dtype == torch.float16 and _is_hot_weight_config(a, n, k):
            w = self.weight.contiguous()
            b4, b8, sb, sb8, wq = _ops.quant_weight_w4(w, float(a.w_clip_ratio), int(a.weight_channel_group),
                                                        return_fake_quant=True)
            self.weight = wq
            self._packed = (b4, b8, sb, sb8)
            self._packed_key = self._weight_key()
            self._unpackable_key = None
            self._f6 = None
            return
        saved = None
        if a.keeper > 0:
            saved = self.weight[:, -a.keeper:].clone().contiguous()
            kp = getattr(a, "keeper_precision", 0)
            if kp == 1:
                saved = fake_quantize_quarter_E5M2(saved)
            elif kp == 2:
                saved = fake_quantize_quarter_E4M3(saved)
            elif kp == 3:
                saved = quantize_tensor(saved, n_bits=8, group_size=0, tiling=0, sym=True)
            self.weight[:, -a.keeper:] = 0
        self.weight = quantize_tensor_channel_group(self.weight.clone(), n_bits=a.wbits, exponential=a.exponential,
                                                    sym=a.w_sym, group_size=a.weight_group_size,
                                                    channel_group=a.weight_channel_group, clip_ratio=a.w_clip_ratio,
                                                    tiling=a.tiling, quant_type=a.quant_type)
        if saved is not None:
            self.weight[:, -a.keeper:] = saved
        self._packed = None
        self._unpackable_key = None
        self._f6 = None

    def reorder(self, in_reorder_index, out_reorder_index=None):
        """reference qLinearLayer.py:80-86."""
        if self.args.reorder is True:
            self.weight = torch.index_select(self.weight, 1, in_reorder_index.to(self.weight.device))
            if out_reorder_index is not None:
                self.weight = torch.index_select(self.weight, 0, out_reorder_index.to(self.weight.device))
            self._packed = None
            self._unpackable_key = None
            self._f6 = None
        return
