// Shared device/host helpers for libatom_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#include "../../include/atom_hip.h"

// Tuning overrides.  The product library takes NO behaviour from the environment (include/atom_hip.h: "no global state"):
// ATOM_TUNE(name, default) is the default, and the string does not even reach the binary.  The tools build (make tools,
// -DATOM_TOOLS) reads the override from the environment variable `name` at every call (a probe may change it between launches).
#ifdef ATOM_TOOLS
#include <cstdlib>
#define ATOM_TUNE(name, dflt) ([]() -> int { const char *e = getenv(name); return e ? atoi(e) : (dflt); }())
#else
#define ATOM_TUNE(name, dflt) (dflt)
#endif

namespace atom {

constexpr int kGroup = 128;
constexpr int kKeeper = 128;

typedef _Float16 half_t;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

// Make a float opaque to the optimiser so that fptrunc(fdiv(fpext a, fpext b)) is NOT narrowed to a
// half-precision divide: the spec is "one correctly rounded FP32 op, then round to FP16".
__device__ __forceinline__ float opaque(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

__device__ __forceinline__ float h2f(half_t h) { return (float)h; }
// FP32 -> FP16 (RNE) of a value that has ALREADY been rounded to FP32.  The opaque() matters: without it LLVM
// fuses fptrunc(fmul/fma) into v_fma_mixlo_f16, which rounds the exact product ONCE to half, while the
// specification (torch/numpy "opmath") is one FP32 op followed by a second rounding to FP16 (measured on gfx950:
// 3.740234375h * 0.9f -> 0x42BB fused vs 0x42BC double-rounded).
__device__ __forceinline__ half_t f2h(float f) { return (half_t)opaque(f); }
__device__ __forceinline__ float round_h(float f) { return (float)(half_t)opaque(f); }

// Offset (halves) of row r's first replica in the reference A-scale layout (Reorder.cuh:39-44).
__host__ __device__ __forceinline__ int ref_scale_index(int r) {
  return (r >> 4) * 64 + (r & 7) * 8 + ((r >> 3) & 1);
}

__host__ __device__ __forceinline__ int64_t ref_scale_size(int64_t x) {
  return x / 16 * 64 + 64 - (1 - (x % 16) / 8) * (8 - (x % 8)) * 8;
}

// Outputs of the fused gate/up epilogue (gemm_w4a4_f6.hip, q kernel with GU != 0): quant(silu(gate) * up) per token and
// 128-feature group, written as the next GEMM's F6 activation operand
struct GateUpOut {
  uint8_t *o6;       // F6 activation buffer [N_inter / 128 - 1][o6_rows][104]
  int64_t o6_rows;   // atom_f6_rows(M)
  int8_t *o8;        // keeper codes [M, 128]
  half_t *s8, *s4;   // keeper scales, group scales (layout per ref_layout; ld4 halves between groups)
  int64_t ld4;
  half_t *xq;        // optional: the fake-quantised activation fp16 [M, N_inter]
  float clip;
  int ref_layout;
};

struct GemmParams {
  const uint8_t *A4, *B4;
  const half_t *sA, *sB;
  const uint8_t *A8, *B8;
  const half_t *sA8, *sB8;
  half_t *D;
  uint8_t *D4;      // o4 epilogue: packed u4 [M, N/2]
  half_t *Dsz;      // o4 epilogue: (scale, zero) [M, N/128, 2]
  float *ws;        // split-K: FP32 partial sums [splits, M, N]
  int splits;       // split-K factor (1 = none)
  int M, N;
  int K4h;          // packed bytes per row of A4/B4 = K4/2
  int G;            // int4 groups
  int ref_layout;
  int a_wide;       // A4 is the wide activation format: int8 [M, K4] = code*16, even/odd de-interleaved per 32 channels
  int64_t ldA;      // halves between groups of sA
  int64_t f6_rows_a, f6_rows_b;   // F6 operand format: padded rows per group of A4 / B4 (gemm_w4a4_f6.hip)
  int o4_ref;                     // ATOM_O4_REF_EXTREMA: the u4 epilogue with the reference code's |x| extrema and 4-bit wrap
  int b_pairs;                    // ATOM_B_SCALE_PAIRS: the caller asserts sB[g][2 j] == sB[g][2 j + 1] (weight_channel_group = 2)
  GateUpOut gu;                   // fused gate/up epilogue (launch_gemm_f6_gateup only)
  const float *sB32;              // ATOM_SB_F32: weight scales float32 [G][f6_rows_b] (then sB is not read by the 256x256 kernel)
  // atom_gemm_w4a4_multi (decode batches): the N output features are nseg segments of seg_n, each with its own [M, seg_n] output
  void *seg_out[3];               // fp16, or float32 where the segment's bit of seg_f32 is set
  const half_t *seg_add;          // optional fp16 [M, seg_n] added to segment 0's fp16 output (the residual stream)
  int seg_n;
  unsigned seg_f32;
  // atom_gemm_w4a4_multi_q (decode, one or two tokens): the activation operand is produced inside the launch from the fp16 input of
  // the quantiser that precedes the GEMM (q_op != 0); A4 / sA / A8 / sA8 are not read
  int q_op;                       // 1 reorder, 2 rmsnorm + reorder, 3 residual add + rmsnorm + reorder, 4 silu(x) * x2
  const half_t *q_x, *q_x2;       // [M, K_total] input; q_x2: rmsnorm weight [K_total] (2, 3) or the second factor [M, K_total] (4)
  const half_t *q_res;            // 3: residual [M, K_total]
  half_t *q_res_out;              // 3: x + residual [M, K_total] (written by workgroup 0; may alias q_res)
  const int16_t *q_idx;           // reorder index [K_total] or null (1, 2, 3)
  float q_eps, q_clip;
  int q_roles;                    // gemvq_w4a4.hip: quantiser waves / streamer waves (set by its launcher)
  const float *q_part;            // 5 (gemvq_w4a4.hip): the decode attention's split partial states [M][heads][q_splits][130] instead of q_x
  int q_splits;
};

int launch_gemm_v2(const GemmParams &p, int ns, hipStream_t s);   // gemm_w4a4_v2.hip
int launch_gemm_v2_o4(const GemmParams &p, hipStream_t s);         // gemm_w4a4_v2.hip, u4 epilogue
int launch_gemm_v3(const GemmParams &p, int cfg, hipStream_t s);   // gemm_w4a4_v3.hip (templated geometry)
int launch_gemv(const GemmParams &p, hipStream_t s);               // gemv_w4a4.hip (M <= 16)
#ifdef ATOM_TOOLS
constexpr int kGemvMaxTokens = 8;                                  // (tuning builds carry the 4- and 8-token instances too)
#else
constexpr int kGemvMaxTokens = 2;                                  // the few-token dot-product kernel (gemv_w4a4.hip gemv1_w4a4_kernel)
#endif
int launch_gemv1(const GemmParams &p, hipStream_t s);              // ... M <= kGemvMaxTokens: fp16 output
int launch_gemv1_f32(const GemmParams &p, hipStream_t s);          // ... FP32 sums into p.ws
int launch_gemv1_multi(const GemmParams &p, hipStream_t s);        // ... segmented outputs (p.seg_*)
int launch_gemm_skinny(const GemmParams &p, hipStream_t s);        // gemm_w4a4_skinny.hip (decode batches, M <= 256)
int launch_gemm_skinny_f32(const GemmParams &p, hipStream_t s);    // ... FP32 sums into p.ws, no final rounding
int launch_gemm_skinny_o4(const GemmParams &p, hipStream_t s);     // ... + the u4 epilogue launch
int launch_gemm_skinny_multi(const GemmParams &p, hipStream_t s);  // ... segmented outputs (p.seg_*): q/k/v, gate/up, down + residual
bool skinny_q_fits(int q_op, int64_t M, int64_t K_total);             // ... the shapes its quantiser-in-front variant takes
int launch_gemm_skinny_multi_q(const GemmParams &p, hipStream_t s);   // ... + the preceding quantiser inside the launch (p.q_*)
bool gemvq_fits(int q_op, int64_t M, int64_t N, int64_t K_total);     // gemvq_w4a4.hip: one or two tokens, the quantiser in front of the dot-product kernel
bool gemvq_merge_fits(int64_t M, int64_t N, int64_t K_total, int splits);   // ... its merge form (q_op 5)
int launch_gemvq_multi_q(const GemmParams &p, hipStream_t s);         // ... (p.q_*, p.seg_*): one quantiser per CU, gemv1's summation order
int launch_gemm_f6(const GemmParams &p, int cfg, hipStream_t s);   // gemm_w4a4_f6.hip (BF6 operands on the block-scaled MFMA)
int launch_gemm_f6_mid(const GemmParams &p, hipStream_t s);        // gemm_w4a4_mid.hip: the same for BF6 operands (ATOM_AB_F6 | ATOM_B_F6S)
int launch_gemm_mid(const GemmParams &p, hipStream_t s);           // gemm_w4a4_mid.hip (packed operands, mid-size batches: 64x64 tiles, deep LDS ring)
int launch_gemm_f6_gateup(const GemmParams &p, int sim, hipStream_t s);   // ... 256x256 kernel + fused SiLU x up -> quant epilogue

// quant_kernels.hip: packed operand (+ scales) -> F6 buffer [G][round_up(rows, 256)][104]
int launch_repack_f6(const uint8_t *src, int64_t rows, int K4h, int G, const half_t *scale, int64_t ld, int ref_layout,
                     uint8_t *out, hipStream_t s);
int launch_repack_f6_pair(const uint8_t *A4, int64_t M, const half_t *sA, int64_t ldA, int ref_layout, uint8_t *outA,
                          const half_t *sB, float *sB32,   // weight scales fp16 [G, N] -> float32 [G][Npad] (both or neither)
                          const uint8_t *B4, int64_t N, uint8_t *outB, int K4h, int G, hipStream_t s);

// ... checkers of the two caller assertions (ATOM_B_SCALE_PAIRS, ATOM_WS_WEIGHT_CACHED): violations are ADDED to *n_bad (device)
int launch_check_scale_pairs(const half_t *sB, int64_t G, int64_t N, int32_t *n_bad, hipStream_t s);
int launch_verify_weight_f6s(const uint8_t *B4, const half_t *sB, int64_t N, int K4h, int G, const uint8_t *have, int32_t *n_bad, hipStream_t s);

// LDS-DMA (global_load_lds_*: global memory -> LDS at M0 + lane * size, no VGPR round trip), issued through inline asm and
// NOT through __builtin_amdgcn_global_load_lds: with the builtin in a loop the compiler's wait-count pass treats the DMA as a
// pending FLAT access and emits `s_waitcnt lgkmcnt(0)` -- a full drain -- in front of EVERY later LDS-read consumer instead of
// the counted wait, which serialises the fragment prefetch of the GEMM kernels (measured on a reduced case: lgkmcnt(2)/(1)
// with the asm form, lgkmcnt(0) with the builtin).  The callers order the DMA themselves (s_waitcnt vmcnt + s_barrier before
// the landed bytes are read); `lds_dst` must be wave-uniform; nothing else in these kernels uses M0.
template <int BYTES>
__device__ __forceinline__ void lds_dma(const void *gsrc, const void *lds_dst) {
  static_assert(BYTES == 16 || BYTES == 4 || BYTES == 2, "LDS-DMA piece size");
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char *)lds_dst);
  if constexpr (BYTES == 16) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(gsrc) : "memory");
  else if constexpr (BYTES == 4) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(m0v), "v"(gsrc) : "memory");
  else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_ushort %1, off" ::"s"(m0v), "v"(gsrc) : "memory");
}

// The same with the LDS byte address as an integer.  lds_addr() of a stage slot once, then integer offsets per piece: the generic -> LDS
// cast carries a null check (64-bit add, compare, select: 4 SALU), which the compiler keeps per piece when every piece casts its own
// pointer (125 -> 97 SALU per K step of the 128x128 F6 kernel).
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p; }
template <int BYTES>
__device__ __forceinline__ void lds_dma_at(const void *gsrc, unsigned lds_byte_addr) {
  static_assert(BYTES == 16 || BYTES == 4 || BYTES == 2, "LDS-DMA piece size");
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  if constexpr (BYTES == 16) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(gsrc) : "memory");
  else if constexpr (BYTES == 4) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(m0v), "v"(gsrc) : "memory");
  else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_ushort %1, off" ::"s"(m0v), "v"(gsrc) : "memory");
}

// ... and with the global address as a wave-uniform base (SGPR pair) + a 32-bit lane offset: no 64-bit VALU add per piece.
template <int BYTES>
__device__ __forceinline__ void lds_dma_sv(const void *sbase, unsigned voff, unsigned lds_byte_addr) {
  static_assert(BYTES == 16 || BYTES == 4 || BYTES == 2, "LDS-DMA piece size");
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  if constexpr (BYTES == 16) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory");
  else if constexpr (BYTES == 4) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory");
  else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_ushort %1, %2" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory");
}

// u4 code of one value under the _o4 epilogue.  REF = the reference code's arithmetic (no clamp, low 4 bits of the int8 cast;
// |result| beyond int8 saturates), else clamp to [0, 15] and 0 for an all-equal group.
template <bool REF>
__device__ __forceinline__ unsigned o4_code(float x, float zero, float rs, float scale) {
  const float t = (x + zero) * rs;
  float tr = truncf(t);
  if (fabsf(t - tr) >= 0.5f) tr += copysignf(1.0f, t);     // roundf: half away from zero
  if constexpr (REF) {
    tr = fminf(fmaxf(tr, -128.f), 127.f);                  // (NaN for a 0 / 0 group -> fmaxf picks -128 -> code 0)
    return (unsigned)(int)tr & 0xFu;
  } else {
    tr = fminf(fmaxf(tr, 0.f), 15.f);
    if (scale == 0.f) tr = 0.f;
    return (unsigned)(int)tr;
  }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// v_cvt_scalef32_2xpk16_bf6_f32 (32 floats -> 32 BF6 fields, the two sources interleaved) through inline asm with an EARLY-CLOBBER
// destination.  With the builtin, hipcc (ROCm 7.2) lets the register allocator place the 6-register result inside the 16-register
// sources at an offset (v[2:7] <- v[0:15]), and the multi-pass instruction then overwrites source elements it has not read yet: a
// re-compile of silu_quant2_kernel turned fields 4.. of some rows into 0 (caught by tests/test_gpu_quant.py).  Result and source at
// the SAME base register are fine; any other overlap is not, and only the constraint rules it out.  (repack_f6_kernel keeps the builtin:
// the asm form doubles its time at 2048 x 13824 x 5120; tests/test_abi_cpu.py disassembles the library and fails on any offset overlap.)
typedef float v16f_t __attribute__((ext_vector_type(16)));
typedef unsigned v6u_t __attribute__((ext_vector_type(6)));
__device__ __forceinline__ v6u_t cvt_2xpk16_bf6(v16f_t a, v16f_t b) {
  v6u_t r;
  asm("v_cvt_scalef32_2xpk16_bf6_f32 %0, %1, %2, 1.0" : "=&v"(r) : "v"(a), "v"(b));
  return r;
}

// Sum over the 64 lanes in the butterfly order xor 32, 16, 8, 4, 2, 1 (every lane ends with the total), without the LDS pipeline:
// v_permlane32_swap / v_permlane16_swap put x[i] and x[i ^ 32] (x[i ^ 16]) side by side in every lane; from then on the partial sums
// repeat with period 16 (8, 4, 2) over the lanes, so the lane i ^ k a stage needs holds the same value as lane (i + k) mod 16 of the
// row: v_add_f32 with the DPP row rotation.  Same additions, same order as six ds_bpermute round trips (the RMSNorm sum of squares, oracle.sumsq_tree; the decode GEMV's final sum).
__device__ __forceinline__ float wave_sum_butterfly(float x) {
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x128, 0xF, 0xF, true));   // row_ror:8
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x124, 0xF, 0xF, true));   // row_ror:4
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x122, 0xF, 0xF, true));   // row_ror:2
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x121, 0xF, 0xF, true));   // row_ror:1
  return x;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device: done once per device and kernel (`done` is a
// per-kernel bit mask of device ids; setting the attribute twice is harmless, so a relaxed race costs one extra call).
inline int ensure_max_lds(const void *kernel, int bytes, std::atomic<uint64_t> &done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ATOM_ERR_LAUNCH;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return ATOM_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return ATOM_ERR_LAUNCH;
  done.fetch_or(bit, std::memory_order_release);
  return ATOM_OK;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? ATOM_OK : ATOM_ERR_LAUNCH; }

}  // namespace atom
