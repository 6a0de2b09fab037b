/*
 * atom_hip.h -- C ABI of libatom_hip.so: MI355X (gfx950) implementation of Atom's W4A4
 * mixed-precision GEMM hot path.
 *
 * Every entry point replaces one host launcher of the reference (file:line relative to
 * /root/reference).  Conventions, all entry points:
 *   - plain device pointers + sizes, no framework types; caller owns every buffer; nothing is
 *     allocated, nothing is synchronised; the launch is enqueued on `stream` (a hipStream_t passed
 *     as void*; NULL = the null stream).  Re-entrant and thread-safe (no global state).
 *   - returns ATOM_OK (0) or a negative ATOM_ERR_* code; shapes/alignment are validated BEFORE any
 *     launch (the reference validates nothing: e2e/punica-atom/punica/ops/csrc/punica_ops.cc:73-80,
 *     211-262).  atom_strerror() names a code.
 *   - group size 128 and keeper (INT8 outlier columns) 128 are the only supported values, as in the
 *     reference (GROUP_SIZE/KEEPER macros, kernels/include/GEMM/Dense_layer_gemm_i4_o16.cuh:35,54).
 *
 * Data formats (identical to the reference unless noted):
 *   A4 / B4   uint8 [rows, K4/2]   two's-complement int4, element 2j in the LOW nibble, 2j+1 in the
 *                                   HIGH nibble (PackInt4, kernels/include/Reorder/Reorder.cuh:16-19)
 *   A8 / B8   int8  [rows, 128]    keeper columns (the LAST 128 reordered channels)
 *   sB        fp16  [G, N]         weight scales, G = K4/128 (Dense_layer_gemm_i4_o16.cuh:497)
 *   sB8       fp16  [N]
 *   sA, sA8   fp16  activation scales, one of two layouts selected by `scale_layout`:
 *       ATOM_SCALE_LAYOUT_REF   (0): the reference's ldmatrix-replicated layout -- row r of group g
 *                                    at g*atom_scale_size(M,0) + scale_index(r) + 2j, j=0..3
 *                                    (Reorder.cuh:39-50,137-156)
 *       ATOM_SCALE_LAYOUT_PLAIN (1): [G, M] row-major (sA8: [M]); the native layout.
 *   K_total = K4 + 128 (the e2e convention, e2e/.../GEMM/DenseLayerGEMM_i4.cu:764).
 */
#ifndef ATOM_HIP_H_
#define ATOM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ATOM_OK 0
#define ATOM_ERR_INVALID_ARG (-22)  /* null pointer, bad enum */
#define ATOM_ERR_SHAPE (-33)        /* unsupported M/N/K/hidden/group/keeper */
#define ATOM_ERR_ALIGN (-14)        /* pointer not 16-byte aligned */
#define ATOM_ERR_LAUNCH (-5)        /* hipGetLastError() != hipSuccess after the launch */

#define ATOM_SCALE_LAYOUT_REF 0
#define ATOM_SCALE_LAYOUT_PLAIN 1

/* Quantisation arithmetic of the three activation ops */
#define ATOM_QUANT_KERNEL 0 /* the reference CUDA kernels: FP32, q = round_half_away(x*(1/s)), s = amax*clip/qmax
                               (Reorder.cuh:137-178).  clip = 1.0 reproduces them exactly. */
#define ATOM_QUANT_SIM 1    /* the reference simulated path: FP16 opmath, amax.clamp(1e-5)*clip, q = rint(x/s)
                               (model/quant.py:141-142,166-172,181).  clip = args.a_clip_ratio. */

/*
 * Native activation format (no reference counterpart): OR ATOM_QUANT_WIDE_CODES into `quant_mode` of the three
 * activation ops and ATOM_A_WIDE into `scale_layout` of atom_gemm_w4a4_f16 / _ws.  o_norms / A4 is then
 *   int8 [M, K4]   byte[m, 32c + 16p + j] = 16 * code[m, 32c + 2j + p]   (c = 32-channel block, p = parity, j < 16)
 * i.e. the INT4 codes already widened to the INT8 MFMA operand (x16, the factor 1/256 is folded into the scales as for
 * packed operands) with the even / odd channels of every 32-channel block de-interleaved -- exactly the two operands the
 * GEMM otherwise makes from one packed 16-byte chunk with 3 VALU instructions per MFMA.  Costs K4/2 more bytes per
 * token in HBM, transient.  Same results bit for bit.  Not accepted by atom_gemm_w4a4_o4; always runs the tile
 * kernels (use the packed format for decode batches, M <= 256: they have their own kernels).
 */
#define ATOM_QUANT_WIDE_CODES 0x100
#define ATOM_A_WIDE 0x100

/*
 * Native "F6" operand format of the block-scaled-MFMA prefill kernel (gemm_w4a4_f6.hip; no reference counterpart): every
 * INT4 code is a BF6 (E3M2) number, and v_mfma_scale_f32_{16x16x128,32x32x64}_f8f6f4 with unit block scales multiply them exactly
 * at the FP4 rate.  OR ATOM_AB_F6 into `scale_layout` of atom_gemm_w4a4_f16: BOTH A4 and B4 are then
 *   uint8 [G][atom_f6_rows(rows)][104]   group-major; bytes 0..95 = the 128 codes of (row, group) as a little-endian
 *                                         stream of 6-bit BF6 fields; A4 only: bytes 96..97 = the fp16 scale of (row,
 *                                         group), 98..99 zero, 100..103 = the same scale as fp32 (the GEMM reads token
 *                                         scales from the row, `sA` is ignored); B4: bytes 96..103 zero
 * with atom_f6_rows(rows) = rows rounded up to 256 (pad rows: any bytes).  sB / keeper operands and scales as before.  OR
 * ATOM_B_F6S in as well when the B4 buffer is followed by the weight scales as float32 [G][atom_f6_rows(N)] (pad columns
 * zero; atom_repack_weight_f6s writes both, offline with the weight): the 256x256 kernel then streams them straight into its
 * de-quantisation registers (no fp16 -> fp32 conversions in the K loop); `sB` (fp16) is still required -- the other tile
 * geometries read it.
 * Activations: ATOM_QUANT_F6_CODES in `quant_mode` of the three activation ops (o_norms = that buffer; norm_scales is
 * still written).  Weights: atom_repack_weight_f6 / atom_repack_weight_f6s.  M, N >= 1 as usual; tile geometries picked by shape, all
 * with the K steps summed in order -- results bit-identical to the INT8 kernels --: 64x64 / 128x64 tiles over the whole K range on a
 * deep LDS ring while a shape has at most 512 tiles of 64x64 (round 5: 256x4096x4096 11.0 us, 512x4096x4096 14.7), 256x256 where
 * those fill the chip (4096^3: 48.5-50 us), 256x128 and 128x128 (4 waves) between -- EXCEPT shapes beyond 512 tiles of 64x64 with at
 * most 256 tiles of 128x128 (and >= 8 K steps; 1024x4096x4096: 23 us): their 128x128 tiles are shared by two groups of 4 waves that
 * split the G + 1 K steps (the G int4 groups, then the keeper) at (G + 1) / 2: each half is summed in order from 0 and
 * D = half(first + second) -- deterministic, same tolerance.  atom_gemm_w4a4_f6_order(M, N, K_total) tells which: 1 = K steps in
 * order, 2 = two halves (4 = four ranges: tuning builds only; 0 = unsupported shape).
 * Ahead of the INT8 kernels from 129 rows up (1.7x at 512-1024 rows, 1.8x at 4096^3).
 */
#define ATOM_QUANT_F6_CODES 0x200
#define ATOM_AB_F6 0x200
int atom_gemm_w4a4_f6_order(int64_t M, int64_t N, int64_t K_total);
#define ATOM_B_F6S 0x400
/* atom_gemm_w4a4_o4 / _o4_ws only: the u4 epilogue exactly as the reference CODE computes it -- its local_max_min takes abs() of
 * both extrema (DenseLayerGEMM_i4_o4.cu:73-80), so scale = (max|x| - min|x|) / 15, zero = -min|x|, and the code is
 * (int8)round((x + zero) / scale) & 0xF (no clamp: negative results wrap, :766-771).  Wrong for any group with negative values
 * (its consumer de-quantises code * scale - zero, flashinfer/quantization.cuh:59-84); the default implements the intended
 * min / max.  Opt-in, for bit-comparisons against a reference run. */
#define ATOM_O4_REF_EXTREMA 0x800
/* atom_gemm_w4a4_f16_ws with PACKED operands on its re-coding route (use (2) below) only: the weight region of `workspace` already
 * holds the F6 form of exactly these B4 / sB (N, K_total) -- written by an earlier call with the same workspace and weight (the
 * weight of a layer is static) -- so only the activation is re-coded: 5 instead of 10 us of re-coding at 4096^3.  The library keeps no
 * state: the CALLER asserts it (atom_amd.ops tracks which weight its workspace holds).  Wrong flag = wrong results, never a fault
 * (the region is inside the workspace either way).  Ignored on every other route.
 * Round 5: with the flag the re-coding route starts at 129 rows (N >= 2048, K >= 1024; at 17 rows where the decode-batch kernel does
 * not take the shape: atom_gemm_w4a4_ws_recodes_cached) -- only the activation is re-coded, and the mid-size-batch kernel behind it runs
 * 129 .. 256 rows in ~11 us at 4096 x 4096 where the decode-batch kernel takes 9.4 .. 15.5, 64 x 13824 x 5120 in 12 us --, so a
 * caller that keeps ONE workspace per weight (a layer's weight is static) fills its weight region once, offline, with
 * atom_repack_weight_f6s(B4, sB, N, K_total, workspace, stream) -- the region's layout is exactly that function's output -- and passes
 * the flag on every call; atom_gemm_w4a4_workspace_bytes() covers those shapes, atom_gemm_w4a4_packed_order(.., 2) names their order. */
#define ATOM_WS_WEIGHT_CACHED 0x1000
/* OR into `scale_layout` of the GEMM entry points: the caller asserts that output channels 2 j and 2 j + 1 share their weight scales,
 * sB[g][2 j] == sB[g][2 j + 1] for every group g (weight_channel_group = 2 of model/qLinearLayer.py:63-70 -- the only form the
 * reference kernel accepts: its dequant applies ONE scale product to both columns of a pair, Dense_layer_gemm_i4_o16.cuh:413-431).
 * The 256x256 block-scaled-MFMA kernel then forms each (token, channel pair, group) scale product once: 6 instead of 8 de-quantisation
 * instructions per MFMA.  Same results bit for bit when the assertion holds; channel 2 j + 1 is de-quantised with channel 2 j's scale
 * when it does not (never a fault).  sB8 -- the keeper's scales are per output channel, model/qLinearLayer.py:59 -- is not covered:
 * the keeper step always forms one product per channel.  Ignored by kernels that do not use it. */
#define ATOM_B_SCALE_PAIRS 0x2000
/* atom_gemm_w4a4_f16_ws only, DEBUG: verify the two assertions above on the device before relying on them -- ATOM_B_SCALE_PAIRS against
 * the values of sB (packed operands), ATOM_WS_WEIGHT_CACHED by re-coding B4 / sB and comparing with the workspace's weight region --
 * and return ATOM_ERR_INVALID_ARG instead of computing wrong numbers when one does not hold.  Needs a workspace of
 * atom_gemm_w4a4_workspace_bytes() (its last 16 bytes serve as the counter before the GEMM uses them); SYNCHRONISES the stream (the one
 * exception to "no synchronisation" in this ABI), so it is not for production calls nor for graph capture.  The reference guards
 * its KV ops the same way at the binding (TORCH_CHECK in punica_ops.cc:18-47); a binding without a debug mode asks the two functions
 * below once per weight instead. */
#define ATOM_WS_VERIFY 0x4000
#define ATOM_F6_PITCH 104

const char *atom_version(void);
const char *atom_strerror(int code);

/* == python scale_size() for layout 0 (punica/ops/__init__.py:137-138, SCALE_SIZE_A Reorder.cuh:50);
 * == rows for layout 1.  Unit: halves. */
size_t atom_scale_size(int64_t rows, int scale_layout);

/*
 * D[M,N] (fp16) = sum_g (A4_g . B4_g^T) * sA[m,g] * sB[g,n]  +  (A8 . B8^T) * sA8[m] * sB8[n]
 * Replaces: DenseLayerGEMM_i4_o16 (kernels/include/GEMM/Dense_layer_gemm_i4_o16.cuh:728-769) and
 *           DenseLayerGEMM_i4<nv_half> (e2e/punica-atom/punica/ops/csrc/GEMM/DenseLayerGEMM_i4.cu:722-791).
 * Integer dot products are exact (INT8 / BF6 MFMA); per group s = sA[m,g] * sB[g,n] -- EXACT in FP32: both are fp16 values, 11-bit
 * significands -- and c = fma(idot, s, c): one rounding per group, groups in order, then the keeper: ONE dot product over its 128
 * columns, de-quantised once the same way -- the reference kernel accumulates both keeper k-steps in int32 before its dequant too
 * (Dense_layer_gemm_i4_o16.cuh:640-691); D = half(c).  This is the reference kernel's own dequant shape -- scale product first
 * (__hmul2, :417), then accu += c_frag * rs_scale (:419-431) -- without its two extra roundings (the FP16 product, the FP32 multiply
 * ahead of the add).  (Rounds 1-4 of this library: t = round_f32(idot * sA), c = fma(t, sB, c).)  The decode kernels
 * (M = 1: dot products along K; 2 <= M <= 256 in the packed format: eight waves own an eighth of the groups each) apply
 * the same per-group arithmetic and add their per-lane / per-wave partial sums in a fixed order: deterministic, within
 * 1 fp16 ulp of the exact value like the tile kernels, but not the same FP32 summation order.
 * Constraints: (K_total-128) % 128 == 0, K_total >= 256, N % 64 == 0, M >= 1, all pointers 16-byte aligned.
 */
int atom_gemm_w4a4_f16(const void *A4, const void *B4, const void *sA, const void *sB,
                       const void *A8, const void *B8, const void *sA8, const void *sB8,
                       void *D, int64_t M, int64_t N, int64_t K_total,
                       int group, int keeper, int scale_layout, void *stream);

/*
 * Same result, with an optional caller-owned scratch buffer.  Two uses: (1) skinny shapes (few output tiles) split
 * the K loop over several workgroups: FP32 partial sums [splits, M, N] are written to `workspace` and reduced, in split
 * order, by a second launch on the same stream; (2) PACKED operands of prefill size (M > 256 -- or > 128 where the decode-batch kernel does not take the shape --,
 * N >= 2048, K >= 1024) are re-coded into the F6 format inside the workspace (one bandwidth-bound launch; the activation alone
 * with ATOM_WS_WEIGHT_CACHED) and multiplied by the BF6-MFMA kernels: 55-60 instead of 89 us at 4096^3, 17.5 (weight cached;
 * 24.3 not) at 512x4096x4096, 26.2 (33.4) at 1024x4096x4096 (round 5); bit-identical to atom_gemm_w4a4_f16 where
 * atom_gemm_w4a4_f6_order(M, N, K_total) == 1, the sum of two / four ordered ranges of the K steps where it is 2 / 4 (see ATOM_AB_F6).
 * atom_gemm_w4a4_workspace_bytes() returns the size that enables it
 * (0 = the shape does not benefit; then, or with a NULL / too small workspace, this is atom_gemm_w4a4_f16).
 * The reference has no counterpart (its 128x128 tile kernel runs every M, bench_dense_layer_gemm_i4_o16.cu:64-69).
 * FP32 summation order differs from the unsplit kernel (per-split partial sums, then their sum): same tolerance.
 */
size_t atom_gemm_w4a4_workspace_bytes(int64_t M, int64_t N, int64_t K_total);
/* 1 when atom_gemm_w4a4_f16_ws takes use (2) for packed operands of this shape (the workspace then starts with the re-coded weight:
 * F6 records + float32 scales, atom_f6_weight_bytes(N, K_total) bytes, whatever M is -- what ATOM_WS_WEIGHT_CACHED refers to) */
int atom_gemm_w4a4_ws_recodes(int64_t M, int64_t N, int64_t K_total);
/* ... the same question for a call that passes ATOM_WS_WEIGHT_CACHED (the weight region holds this weight's BF6 form already, only the
 * activation is re-coded): additionally every shape of 129 rows and more, and from 17 rows the shapes the decode-batch kernel does not
 * take (N >= 2048, K >= 1024 throughout).  What a binding that keeps one workspace per weight asks. */
int atom_gemm_w4a4_ws_recodes_cached(int64_t M, int64_t N, int64_t K_total);
/* Checkers of the caller assertions (asynchronous on `stream`, no synchronisation; the count is a device int32 the caller reads):
 *   atom_check_scale_pairs   *n_bad_dev = number of (group, channel pair) of sB fp16 [G][N] whose two scales differ -- 0 means
 *                            ATOM_B_SCALE_PAIRS may be asserted for this weight (what atom_amd/ops.py:scale_pairs_shared computes in torch)
 *   atom_verify_weight_f6s   *n_bad_dev = number of (row < N, group) whose F6 record or float32 scale in B_f6s (atom_f6_weight_bytes(N,
 *                            K_total) bytes: a workspace's weight region) is not what atom_repack_weight_f6s makes of B4 / sB -- 0 means
 *                            ATOM_WS_WEIGHT_CACHED may be asserted for this (workspace, weight)
 * No reference counterpart (its launcher has neither flag; its bindings check what they can with TORCH_CHECK, punica_ops.cc:18-47). */
int atom_check_scale_pairs(const void *sB, int64_t G, int64_t N, int32_t *n_bad_dev, void *stream);
int atom_verify_weight_f6s(const void *B4, const void *sB, int64_t N, int64_t K_total, const void *B_f6s, int32_t *n_bad_dev, void *stream);
int atom_gemm_w4a4_f16_ws(const void *A4, const void *B4, const void *sA, const void *sB,
                          const void *A8, const void *B8, const void *sA8, const void *sB8,
                          void *D, int64_t M, int64_t N, int64_t K_total,
                          int group, int keeper, int scale_layout,
                          void *workspace, size_t workspace_bytes, void *stream);
/* The FP32 summation order atom_gemm_w4a4_f16 (with_workspace = 0) / atom_gemm_w4a4_f16_ws with a workspace of
 * atom_gemm_w4a4_workspace_bytes() bytes (with_workspace = 1) applies to PACKED (reference-format) operands of this shape -- a host-side
 * function of the shape alone, what the parity tests restate the arithmetic for (oracle/atom_oracle.c):
 *   1        the K steps (int4 groups, then the keeper) in order: the tile kernels -- among them the mid-size-batch kernels
 *            (gemm_w4a4_mid.hip: 64 x 64 / 128 x 64 tiles on a deep LDS ring; the INT8 form for 17 .. 256 rows with 96 .. 256 tiles, the
 *            BF6 form behind the re-coding routes up to 512 tiles of 64 x 64)
 *   2        two ordered ranges of the K steps (the BF6 two-wave-group kernel behind the re-coding routes; see ATOM_AB_F6)
 *   8        the decode-batch kernel: the G + 1 steps dealt to 8 waves in consecutive slices, partial sums added in wave order
 *   64       the one- / two-token dot-product kernel: groups dealt to 16 quad leaders, 64-lane butterfly, keeper last
 *   63       the staged dot-product kernel (M <= 7 with a K too long for the decode-batch kernel): per-lane partial sums
 *   100 + s  split-K over s workgroups through the workspace, partial sums added in split order
 *   0        unsupported shape
 * with_workspace = 2: a workspace AND ATOM_WS_WEIGHT_CACHED (the re-coding route from 129 rows, and from 17 rows for the shapes the
 * decode-batch kernel does not take: atom_gemm_w4a4_ws_recodes_cached). */
int atom_gemm_w4a4_packed_order(int64_t M, int64_t N, int64_t K_total, int with_workspace);

/*
 * Same GEMM; epilogue asymmetric-quantises every 128-wide output group to u4:
 *   scale = (max-min)/15, zero = -min, q = round((x+zero)/scale) & 0xF
 * D_u4 uint8 [M, N/2] (same nibble order), D_scale_zero fp16 [M, N/128, 2] = (scale, zero).
 * Replaces: DenseLayerGEMM_i4_o4 (e2e/.../GEMM/DenseLayerGEMM_i4_o4.cu:808-856, epilogue :704-788).
 * Constraint in addition: N % 128 == 0.
 */
int atom_gemm_w4a4_o4(const void *A4, const void *B4, const void *sA, const void *sB,
                      const void *A8, const void *B8, const void *sA8, const void *sB8,
                      void *D_u4, void *D_scale_zero, int64_t M, int64_t N, int64_t K_total,
                      int group, int keeper, int scale_layout, void *stream);

/*
 * Decode batches: ONE launch for the projections of a decode step that share an activation operand (no reference counterpart: the
 * reference launches dense_layer_gemm_i4_fp16 / _o4 once per projection, punica/models/llama.py:85-87 gate / up, :110-119 q / k / v).
 * B4 / sB / B8 / sB8 hold the `nseg` (1..3) weights concatenated along the output features: B4 [nseg * N_seg, K4/2], sB [G, nseg * N_seg]
 * flat, B8 [nseg * N_seg, 128], sB8 [nseg * N_seg].  Segment s gets its own output out_s [M, N_seg]: fp16, or the FP32 sums where bit s
 * of f32_mask is set (k / v ahead of atom_kv_quant_append_f32).  add0_f16 (optional, fp16 [M, N_seg]): added to segment 0's fp16
 * output as torch adds two half tensors -- half(float(half(c)) + float(add)) -- i.e. the decoder layer's residual add
 * (llama.py:268-275, :289-291) in the projection's own launch.  Per feature the arithmetic and the summation order are those of
 * atom_gemm_w4a4_f32 / of the decode-batch kernel behind atom_gemm_w4a4_f16: bit-identical to the separate launches (+ torch's add).
 * Shapes: atom_gemm_w4a4_multi_fits(M, N_seg, nseg, K_total) (decode batches; N_seg % 16 == 0); packed operands only.
 */
int atom_gemm_w4a4_multi_fits(int64_t M, int64_t N_seg, int nseg, int64_t K_total);
int atom_gemm_w4a4_multi(const void *A4, const void *B4, const void *sA, const void *sB, const void *A8, const void *B8,
                         const void *sA8, const void *sB8, void *out0, void *out1, void *out2, unsigned f32_mask,
                         const void *add0_f16, int64_t M, int64_t N_seg, int nseg, int64_t K_total, int group, int keeper,
                         int scale_layout, void *stream);

/*
 * atom_gemm_w4a4_multi with the quantiser that PRECEDES the GEMM inside the launch: one or two tokens of a decode step
 * (atom_gemm_w4a4_multi_q_fits).  The reference runs the two as separate ops -- punica/models/llama.py:259-263 input_layernorm
 * (rmsnorm_fp16_i4) -> :110-119 q / k / v; :176 reorder_fp16_i4 -> o_proj; :266-282 residual add + post_attention_layernorm -> :85
 * gate / up; :86 activate_fp16_i4 -> :87 down_proj -- and so does this library from three tokens on; at one or two tokens a
 * quantiser launch costs more than its arithmetic (a launch boundary, its own round trip through HBM, 4.4-4.7 us against the GEMM's 5-9), so
 * the GEMM launch quantises the token rows itself: once per CU, in front of the dot-product kernel's feature loop (round 6,
 * csrc/gemvq_w4a4.hip: one workgroup of 16 waves per CU; a Llama-7B decode layer at batch 1, cold, 60.8 -> 52 us with all four
 * quantisers inside).  (Rounds 3-5: in every 16-feature workgroup of the decode-batch kernel -- still the form for shapes the
 * dot-product form does not take.)
 *   q_op   ATOM_Q_REORDER      x [M, K_total] fp16, reorder_index (or NULL)           = atom_reorder_quant_f16
 *          ATOM_Q_RMSNORM      + x2 = the RMSNorm weight [K_total], eps               = atom_rmsnorm_reorder_quant_f16
 *          ATOM_Q_ADD_RMSNORM  + residual, residual_out [M, K_total] (x + residual)   = atom_add_rmsnorm_reorder_quant_f16
 *                              (NOT in place: every workgroup reads x and residual, workgroup 0 writes residual_out)
 *          ATOM_Q_SILU_MUL     x2 = the second factor [M, K_total]; no reorder index  = atom_silu_mul_quant_f16
 * in the kernel-flavoured arithmetic (quant_mode 0), clip as there.  Outputs, segments, f32_mask, add0_f16 as atom_gemm_w4a4_multi.
 * The quantised operand is bit-identical to the quantiser op's, and the sums are formed in the order atom_gemm_w4a4_multi uses for the
 * shape (atom_gemm_w4a4_packed_order: 64, the dot-product kernel's, for one token and for two with K_total > 4096 -- the quantiser then
 * sits in front of that kernel; 8, the decode-batch kernel's, for two tokens with K_total <= 4096): the outputs are bit-identical to the
 * quantiser op followed by atom_gemm_w4a4_multi (tests/test_gpu_e2e.py).  (Rounds 3-5 always summed in the decode-batch kernel's order
 * and were one fp16 ulp off where the separate call took the dot-product kernel.)
 */
#define ATOM_Q_REORDER 1
#define ATOM_Q_RMSNORM 2
#define ATOM_Q_ADD_RMSNORM 3
#define ATOM_Q_SILU_MUL 4
/* The same launch with the decode attention's KV-split merge in front of the reorder quantiser (round 6): `partials_f32` = the
 * workspace atom_batch_decode_append_i4 / atom_batch_decode_i4 filled when called with o == NULL -- float [M][heads][splits][130] =
 * 128 un-normalised values, the running maximum m (base 2) and the denominator d per (token, head, split), heads = K_total / 128 --
 * merged as the decode op's own merge launch does (out = sum_s o_s 2^(m_s - M) / sum_s d_s 2^(m_s - M), splits in order), rounded to
 * fp16, then reordered, quantised and multiplied as ATOM_Q_REORDER: bit-identical to batch_decode (merged) -> atom_gemm_w4a4_multi_q
 * (ATOM_Q_REORDER), two launch boundaries of a decode step fewer (the reference: punica/models/llama.py:168-196 batch_decode ->
 * reorder_fp16_i4 -> o_proj).  2 .. 16 splits, one or two tokens, M x K_total <= 8192: atom_gemm_w4a4_multi_merge_q_fits. */
int atom_gemm_w4a4_multi_merge_q_fits(int64_t M, int64_t N_seg, int nseg, int64_t K_total, int splits);
int atom_gemm_w4a4_multi_merge_q(const void *partials_f32, int splits, const int16_t *reorder_index, float clip, const void *B4, const void *sB,
                                 const void *B8, const void *sB8, void *out0, void *out1, void *out2, unsigned f32_mask,
                                 const void *add0_f16, int64_t M, int64_t N_seg, int nseg, int64_t K_total, int group, int keeper,
                                 void *stream);
/* 1 when atom_gemm_w4a4_multi_q takes (q_op, M, N_seg, nseg, K_total): the launcher's own predicate -- one or two tokens, K_total <=
 * 12,288 (three 4-channel quantiser tasks per thread of the 1024 and token row), for the three ops that stage fp16 rows and the norm
 * weight in LDS M x K_total <= 16,384, and a shape atom_gemm_w4a4_multi takes. */
int atom_gemm_w4a4_multi_q_fits(int q_op, int64_t M, int64_t N_seg, int nseg, int64_t K_total);
int atom_gemm_w4a4_multi_q(int q_op, const void *x, const void *x2, const void *residual, void *residual_out,
                           const int16_t *reorder_index, float eps, float clip, const void *B4, const void *sB, const void *B8,
                           const void *sB8, void *out0, void *out1, void *out2, unsigned f32_mask, const void *add0_f16, int64_t M,
                           int64_t N_seg, int nseg, int64_t K_total, int group, int keeper, void *stream);

/*
 * Same result contract, with an optional caller-owned scratch buffer for DECODE batches (the k/v projections of a
 * serving step; the shapes the decode-batch GEMM takes, M <= 256 at most).  The plain entry point runs the 256-row
 * tile kernel whatever M is (88 us at M = 16, N = K = 4096); with
 * a workspace of atom_gemm_w4a4_o4_workspace_bytes() bytes the weight-streaming decode kernel writes FP32 sums [M, N]
 * there and a second launch applies the u4 epilogue (same arithmetic; the FP32 summation order is the decode kernel's,
 * so a code can differ by one from the tile kernel's where a value sits on a rounding boundary).  0 bytes = the shape
 * does not use it; then, or with a NULL / too small workspace, this is atom_gemm_w4a4_o4.
 */
size_t atom_gemm_w4a4_o4_workspace_bytes(int64_t M, int64_t N, int64_t K_total);
int atom_gemm_w4a4_o4_ws(const void *A4, const void *B4, const void *sA, const void *sB,
                         const void *A8, const void *B8, const void *sA8, const void *sB8,
                         void *D_u4, void *D_scale_zero, int64_t M, int64_t N, int64_t K_total,
                         int group, int keeper, int scale_layout, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same sums WITHOUT the final rounding: D_f32 float [M, N] = the FP32 accumulators the u4 epilogue quantises.
 * Decode batches in the packed format only (the shapes the decode-batch GEMM takes; ATOM_ERR_SHAPE otherwise).  Feeds
 * atom_kv_quant_append_f32 below: k / v projection -> u4 codes -> paged cache in two launches per projection pair
 * instead of five.  No reference counterpart (its _o4 GEMM quantises in its epilogue, the append is a separate kernel).
 */
int atom_gemm_w4a4_f32(const void *A4, const void *B4, const void *sA, const void *sB,
                       const void *A8, const void *B8, const void *sA8, const void *sB8,
                       void *D_f32, int64_t M, int64_t N, int64_t K_total,
                       int group, int keeper, int scale_layout, void *stream);

/*
 * The three fused activation-quantisation ops.  Common outputs (row r, hidden = H, K4 = H-128):
 *   o_outliers     int8  [M, 128]        INT8 codes of the last 128 (reordered) channels
 *   o_norms        uint8 [M, K4/2]       packed INT4 codes of the first K4 channels
 *   outlier_scales fp16  atom_scale_size(M, layout) halves
 *   norm_scales    fp16  [K4/128, atom_scale_size(M, layout)]
 *   xq_f16         fp16  [M, H] or NULL  optional de-quantised tensor (codes*scale in half) == what the
 *                                         reference's quantize_activation_wrapper returns (sim mode)
 * Constraints: H % 128 == 0, 256 <= H <= 16384, reorder_index int16 [H] with values in [0,H); a NULL
 * reorder_index means the identity (input already in reordered channel order).
 */

/* y = index_select(x, -1, reorder_index); quantise.
 * Replaces: run_reorder_fp16_i4<128,4096> (kernels/include/Reorder/Reorder.cuh:205-228);
 * sim mode == index_select + quantize_activation_wrapper (model/qLlamaLayer.py:300-304). */
int atom_reorder_quant_f16(const void *x, const int16_t *reorder_index, int64_t M, int hidden,
                           int quant_mode, float clip, int scale_layout,
                           void *o_outliers, void *o_norms, void *outlier_scales, void *norm_scales,
                           void *xq_f16, void *stream);

/* y = RMSNorm(x)*w; index_select; quantise.
 * Replaces: run_rmsnorm_fp16_i4<128,4096> (kernels/include/RMSNorm/RMSNorm.cuh:255-285);
 * sim mode == QLlamaRMSNorm.forward (model/qLlamaLayer.py:141-151) with HF LlamaRMSNorm numerics. */
int atom_rmsnorm_reorder_quant_f16(const void *x, const void *weight, float eps,
                                   const int16_t *reorder_index, int64_t M, int hidden,
                                   int quant_mode, float clip, int scale_layout,
                                   void *o_outliers, void *o_norms, void *outlier_scales,
                                   void *norm_scales, void *xq_f16, void *stream);

/* NEW: residual_out[m,:] = x[m,:] + residual[m,:] (one fp16 add per element; residual_out may be
 * `residual` itself -- the residual stream of the decoder layer, llama.py:266-282), then exactly
 * atom_rmsnorm_reorder_quant_f16 of that sum.  Replaces the elementwise add kernel + the RMSNorm kernel's re-read of its
 * result (24 -> 16 KiB of HBM traffic per token at hidden 4096, one launch less). */
int atom_add_rmsnorm_reorder_quant_f16(const void *x, const void *residual, void *residual_out, const void *weight,
                                       float eps, const int16_t *reorder_index, int64_t M, int hidden,
                                       int quant_mode, float clip, int scale_layout,
                                       void *o_outliers, void *o_norms, void *outlier_scales,
                                       void *norm_scales, void *xq_f16, void *stream);

/* y = silu(a)*b; quantise (no reorder: the weights are pre-permuted, modelutils_llama.py:33-40).
 * Replaces: run_activate_fp16_i4<128,11008> (kernels/include/Activate/Activate.cuh:194-217);
 * sim mode == act_fn(gate)*up -> act_quant (model/qLlamaLayer.py:345-351). */
int atom_silu_mul_quant_f16(const void *a, const void *b, int64_t M, int hidden,
                            int quant_mode, float clip, int scale_layout,
                            void *o_outliers, void *o_norms, void *outlier_scales, void *norm_scales,
                            void *xq_f16, void *stream);

/*
 * NEW (the bridge the reference lacks): quantise + pack an FP16 weight [N,K] (columns
 * already reordered, keeper = last 128 columns) exactly as QLinearLayer.quant does
 * (model/qLinearLayer.py:42-78, model/quant.py:68-107): INT8 per output row for the keeper columns,
 * INT4 with `channel_group` (1 or 2) adjacent rows sharing a scale per 128-column slice, clip w_clip.
 * Outputs: B4 uint8 [N,K4/2], B8 int8 [N,128], sB fp16 [G,N], sB8 fp16 [N], Wq_f16 fp16 [N,K] or NULL
 * (the fake-quant weight the reference would hold).
 */
int atom_quant_weight_w4(const void *W_f16, int64_t N, int64_t K_total, float w_clip,
                         int channel_group, void *B4, void *B8, void *sB, void *sB8, void *Wq_f16,
                         void *stream);

/*
 * NEW (the bridge for GPTQ-written weights): pack a weight that is ALREADY fake-quantised -- every
 * value half(s*c) with c in [-8,7] sharing s per (`channel_group` rows x 128 columns), last 128 columns c in
 * [-128,127] with one s per row -- by recovering (c, s).  This is what GPTQ leaves in `layer.weight.data`
 * (model/gptq.py:38-39 quantises with an FP32 scale found on the error-compensated weight, :285-287, stores half(q),
 * :331, and discards the scale) and also what QLinearLayer.quant leaves (model/qLinearLayer.py:42-78; reproduced
 * exactly, the scale there is fp16).  For FP32-scale weights sB is the fp16 value that reproduces Wq best:
 * |c*sB - Wq| <= 2^-9 |Wq| + 2^-24 per element.  Outputs as atom_quant_weight_w4.  `bad_blocks` (device int32, may be
 * NULL) receives the number of blocks that are on no such grid; those are re-quantised round-to-nearest (the caller
 * decides whether that is an error -- the Python mirror raises).
 */
int atom_pack_weight_w4(const void *Wq_f16, int64_t N, int64_t K_total, int channel_group,
                        void *B4, void *B8, void *sB, void *sB8, int32_t *bad_blocks, void *stream);

/*
 * INT4 paged KV cache -- the two ops that follow the k/v projections (atom_gemm_w4a4_o4) in the
 * reference's decoder layer (e2e/punica-atom/punica/models/llama.py:196-211).  Layouts are the reference's
 * (punica/utils/kvcache.py:17-26, kernels/include/flashinfer/page.cuh:78-110):
 *   kv_data  uint8 [pages, num_layers, 2, num_heads, page_size, head_dim/2]  u4, element 2j in the low nibble
 *   kv_param fp16  [pages, num_layers, 2, num_heads, page_size, 2]           (scale, zero): value = u4*scale - zero
 *   kv_indptr int32 [batch+1], kv_indices int32 [kv_indptr[batch]], last_page_offset int32 [batch] (1..page_size);
 *   sequence b has (npages_b - 1)*page_size + last_page_offset[b] tokens.
 * head_dim must be 128 (as in the reference, punica_ops.cc:112), page_size a multiple of 16.
 */

/* Copy new tokens' quantised K/V (the d / d_scale outputs of atom_gemm_w4a4_o4, viewed [T, heads, 64] / [T, heads, 2])
 * into the pages.  append_indptr int32 [batch+1] (device): the tokens are the LAST append_indptr[b+1]-append_indptr[b]
 * positions of sequence b (pages already allocated, lengths already updated), total_tokens = append_indptr[batch];
 * append_indptr == NULL: one token per sequence (total_tokens == batch).
 * Replaces: FlashInferInitKvKernel_i4 / FlashInferAppendKvKernel_i4 (punica/ops/csrc/flashinfer_adapter/
 * flashinfer_impl.cuh:48-96; page.cuh:119-227) = punica_ops.cc:122-209 init_kv_i4 / append_kv_i4. */
int atom_kv_append_i4(void *kv_data, void *kv_param, const int32_t *kv_indptr, const int32_t *kv_indices,
                      const int32_t *last_page_offset, const void *k, const void *v, const void *k_param,
                      const void *v_param, const int32_t *append_indptr, int64_t total_tokens, int batch,
                      int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, void *stream);

/* Decode step, fused: quantise the FP32 sums of the k / v projections (atom_gemm_w4a4_f32: float [batch, heads*128] each)
 * per head with the arithmetic of the _o4 epilogue (DenseLayerGEMM_i4_o4.cu:704-788) and write codes + (scale, zero)
 * into the LAST token's slot of each sequence (the one-token-per-sequence form of FlashInferAppendKvKernel_i4,
 * flashinfer_impl.cuh:72-96).  Bit-identical cache contents to atom_gemm_w4a4_o4_ws + atom_kv_append_i4. */
int atom_kv_quant_append_f32(void *kv_data, void *kv_param, const int32_t *kv_indptr, const int32_t *kv_indices,
                             const int32_t *last_page_offset, const void *k_f32, const void *v_f32, int batch,
                             int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, void *stream);

/* o[b,h,:] (fp16) = softmax_j( <RoPE(q[b,h], len_b-1), RoPE(dequant K[b,h,j], j)> / sqrt(128) ) . dequant V[b,h,j].
 * Replaces: FlashInferBatchDecodeKernel_i4 (flashinfer_impl.cuh:9-46; decode.cuh:480-676) = punica_ops.cc:82-120
 * batch_decode_i4 (rope_theta 1e4, rope_scale 1 there).  max_pages_per_seq (host-side upper bound of npages_b, 0 =
 * unknown) lets long sequences at small batch split their KV range over several waves: FP32 partial states go to
 * `workspace` (atom_batch_decode_i4_workspace_bytes; NULL / too small = no split) and a second launch merges them.  Where
 * batch x num_heads >= 256 and the split count divides 12 (round 6) the splits are waves of one workgroup instead and merge in LDS:
 * one launch, the workspace is not touched, the same bits. */
size_t atom_batch_decode_i4_workspace_bytes(int batch, int num_heads, int page_size, int max_pages_per_seq);
/* ... how many PARTIAL STATES per (sequence, head) the call produces for these arguments (1 = none: no workspace use).  At small batches
 * the waves that share a pair's KV range come in workgroups of four that leave one partial state each (round 6), so this is the
 * number of such workgroups, not of waves.  With o == NULL and a
 * split count >= 2 the two decode entry points leave the partial states in the workspace UN-merged, as float [batch][heads][splits][130]
 * (128 un-normalised values, running maximum, denominator), for a consumer that merges them itself: atom_gemm_w4a4_multi_merge_q. */
int atom_batch_decode_i4_splits(int batch, int num_heads, int page_size, int max_pages_per_seq);
int atom_batch_decode_i4(void *o, const void *q, const void *kv_data, const void *kv_param, const int32_t *kv_indptr,
                         const int32_t *kv_indices, const int32_t *last_page_offset, int batch, int num_layers,
                         int layer_idx, int num_heads, int page_size, int head_dim, float rope_theta, float rope_scale,
                         int max_pages_per_seq, void *workspace, size_t workspace_bytes, void *stream);
/* atom_kv_quant_append_f32 + atom_batch_decode_i4 of a pure decode step in ONE launch (round 6): k_f32 / v_f32 (float [batch, heads*128])
 * are this step's k / v projections; for every (sequence, head) the wave that attends to the last token quantises them (the same _o4
 * arithmetic), writes codes + (scale, zero) into the last token's cache slot and attends to the values it has just written -- the
 * reference's call order punica/models/llama.py:168-196 (append_kv, then batch_decode) with bit-identical cache contents and output,
 * without the launch boundary between the two. */
int atom_batch_decode_append_i4(void *o, const void *q, const void *k_f32, const void *v_f32, void *kv_data, void *kv_param,
                                const int32_t *kv_indptr, const int32_t *kv_indices, const int32_t *last_page_offset, int batch,
                                int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, float rope_theta,
                                float rope_scale, int max_pages_per_seq, void *workspace, size_t workspace_bytes, void *stream);

/*
 * KV-cache fake quantisation of the simulated path: every 128-d head vector of x is quantised
 * asymmetrically to n_bits in FP16 opmath -- scale = ((max - min) * clip).clamp(1e-5) / (2^n - 1), base =
 * round(-min / scale).clamp(0, 2^n - 1), y = (clamp(round(x / scale) + base, 0, 2^n - 1) - base) * scale -- and written
 * back as fp16.  Replaces: quantize_attn_k_wrapper / quantize_attn_v_wrapper (model/quant.py:233-257 ->
 * quantize_tensor(sym=False), :143-145,173-181).  x is a [batch, num_heads, seq_len, 128] view with the given ELEMENT
 * strides (last dim contiguous; the reference feeds the transposed projection output), y is contiguous; y == x is
 * allowed when x is contiguous.
 */
int atom_kv_fake_quant_f16(const void *x, void *y, int64_t batch, int num_heads, int64_t seq_len, int64_t stride_b,
                           int64_t stride_h, int64_t stride_s, int n_bits, float clip, void *stream);

/* Rows per group of an F6 operand buffer: `rows` rounded up to 256 (buffer bytes = G * atom_f6_rows(rows) * 104). */
size_t atom_f6_rows(int64_t rows);

/* Packed INT4 weights B4 [N, K4/2] -> the F6 format (see ATOM_AB_F6): B_f6 uint8 [G][atom_f6_rows(N)][104], pad rows and
 * bytes 96..103 zeroed.  Offline, once per weight. */
int atom_repack_weight_f6(const void *B4, int64_t N, int64_t K_total, void *B_f6, void *stream);

/* Packed INT4 activations A4 [M, K4/2] + their scales sA (per `scale_layout`) -> the F6 activation operand [G][atom_f6_rows(M)][104]
 * with the token scales in the rows (what ATOM_QUANT_F6_CODES makes the quantisers emit directly; this is for callers that hold the
 * reference's packed format, e.g. ahead of atom_gemm_w4a4_silu_mul_quant_f6).  One bandwidth-bound launch. */
int atom_repack_act_f6(const void *A4, const void *sA, int64_t M, int64_t K_total, int scale_layout, void *A_f6, void *stream);

/* The same + the fp16 weight scales sB [G, N] appended as float32 [G][atom_f6_rows(N)] (ATOM_B_F6S): B_f6s holds
 * atom_f6_weight_bytes(N, K_total) = G * atom_f6_rows(N) * (104 + 4) bytes. */
size_t atom_f6_weight_bytes(int64_t N, int64_t K_total);
int atom_repack_weight_f6s(const void *B4, const void *sB, int64_t N, int64_t K_total, void *B_f6s, void *stream);

/*
 * gate_proj + up_proj + act_fn(gate) * up + the per-token group quantiser in ONE launch: what the reference runs as
 * two dense_layer_gemm_i4_fp16 calls and activate_fp16_i4 (punica/models/llama.py:85-87; model/qLlamaLayer.py:345-351).  A_f6 is the
 * F6 activation operand [G][atom_f6_rows(M)][104]; Bgu_f6s is the F6 weight buffer WITH appended fp32 scales
 * (atom_repack_weight_f6s) of the 2 * N_inter fused rows: per block j of 128 features and quarter w = 0..3, gate rows
 * 128 j + 32 w .. + 31 followed by the same 32 up rows; Bgu8 int8 [2 N_inter, 128] and sBgu8 fp16 [2 N_inter] in the same row
 * order.  Outputs = atom_silu_mul_quant_f16's with ATOM_QUANT_F6_CODES, for hidden N_inter: o_outliers int8 [M, 128], o_norms_f6
 * uint8 [N_inter / 128 - 1][atom_f6_rows(M)][104] (the F6 activation operand of down_proj), outlier_scales / norm_scales per
 * `scale_layout`, optional xq fp16 [M, N_inter] (NULL: not written).  quant_mode ATOM_QUANT_SIM / ATOM_QUANT_KERNEL, clip as there.
 * The K steps are summed in order (fp16 GEMM outputs are formed in registers and go through the same arithmetic): bit-identical to
 * the three launches it replaces wherever atom_gemm_w4a4_f6_order(M, N_inter, K_total) == 1 (every shape that fills the chip); for
 * few-tile shapes, where the stand-alone GEMMs add two / four ordered K ranges, same tolerance instead (the module wrappers fuse only
 * in the first case).  Saves writing and re-reading 2 * M * N_inter fp16.  Always the 256x256 geometry: meant for prefill batches (M >= 512).
 * N_inter % 128 == 0, N_inter >= 256; K_total as atom_gemm_w4a4_f16.
 */
int atom_gemm_w4a4_silu_mul_quant_f6(const void *A_f6, const void *Bgu_f6s, const void *A8, const void *Bgu8, const void *sA8,
                                     const void *sBgu8, int64_t M, int64_t N_inter, int64_t K_total, int group, int keeper,
                                     int quant_mode, float clip, int scale_layout, void *o_outliers, void *o_norms_f6,
                                     void *outlier_scales, void *norm_scales, void *xq, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ATOM_HIP_H_ */
