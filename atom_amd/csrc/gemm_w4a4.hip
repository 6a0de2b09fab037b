// W4A4 group-128 mixed-precision GEMM for gfx950 (MI355X, CDNA4).
//
//   D[M,N] = sum_g (A4_g . B4_g^T) sA[m,g] sB[g,n] + (A8 . B8^T) sA8[m] sB8[n]        (fp16 out)
//
// Replaces compute_gemm_imma / DenseLayerGEMM_i4_o16 (reference
// kernels/include/GEMM/Dense_layer_gemm_i4_o16.cuh:436-769).  NOT a translation: the reference is built on
// cp.async rings + ldmatrix + the native INT4 mma.m16n8k64; gfx950 has none of them.  The CDNA4 design:
//
//  * MFMA: v_mfma_i32_32x32x32_i8 (gfx950 has no INT4 MFMA).  int4 -> int8 widening happens ONCE per
//    workgroup while staging into LDS, not per wave: a packed dword v gives (v<<4)&0xF0F0F0F0 (even
//    elements) and v&0xF0F0F0F0 (odd elements) = the int4 values *16 as int8, sign in place, 3 VALU ops
//    per 8 values.  Both operands use the same even/odd split, and a dot product does not care about the
//    order of k, so no re-interleave is needed; the 16*16 = 256 factor is folded into the activation
//    scale (exact power of two).
//  * The MFMA computes the TRANSPOSED tile (weights as the MFMA A operand, activations as B): in the
//    32x32 accumulator layout a lane then owns ONE token row m and 16 output features n, so the
//    per-token scale sA[m] is a lane scalar and the epilogue writes n-contiguous 8-byte words.
//  * Per-group dequant costs 2 VALU ops per accumulator element instead of 3 (cvt+mul+fma): the integer
//    accumulator of each group starts at the bit pattern of 1.5*2^23 (passed as the MFMA's C operand),
//    so after the group's MFMAs the register, READ AS A FLOAT, is exactly 12582912 + idot.  Then
//        t = fma(acc_as_float, sA', -(12582912*sA'))   == round_f32(idot * sA')   (12582912*sA' is exact)
//        c = fma(t, sB[n], c)
//    |idot*256| <= 2^21 < 2^22 keeps the mantissa trick exact for both the int4 and the int8 groups.
//  * LDS tiles hold int8 rows of 128 bytes (one quantisation group of K per step), 16-byte slots XOR-
//    swizzled by (row>>1)&7 so that ds_read_b128 fragment reads are bank-conflict free; double buffered,
//    one barrier per K-group; global->register prefetch of the next group is issued before the MFMAs of
//    the current one (T14 "issue early / write late").
//  * The 128 INT8 keeper columns run as two extra 64-wide steps through the same pipeline.
#include "common.h"
#include <cstdlib>

namespace atom {

constexpr float kMagic = 12582912.0f;          // 1.5 * 2^23
constexpr int kMagicBits = 0x4B400000;

template <int BM_, int BN_, int WGM_, int WGN_>
struct GemmCfg {
  static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_;
  static constexpr int NW = WGM * WGN, NT = NW * 64;
  static constexpr int WM = BM / WGM, WN = BN / WGN;
  static constexpr int TM = WM / 32, TN = WN / 32;
  static constexpr int ROWS = BM + BN;
  static constexpr int NCH = ROWS * 4 / NT;                 // 16-byte chunks per thread per step
  static constexpr int TILE_BYTES = ROWS * 128;
  static constexpr int STAGE_BYTES = TILE_BYTES + ROWS * 4;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
  static_assert(ROWS * 4 % NT == 0, "chunks must divide evenly");
  static_assert(ROWS <= NT, "one scale per thread");
  static_assert(BN % 16 == 0 && BM % 16 == 0, "swizzle period");
};

template <class C>
struct GemmState {
  v4u pre[C::NCH];
  float pre_scale;
};

// ---- global -> registers (issued early; consumed by stage_store after the MFMAs) -------------------
// Branch-free: the operand kind of chunk i is a compile-time property (BN is a multiple of NT/4 rows),
// int4-vs-keeper is a wave-uniform select of (base, row stride, k offset), and rows past M / N are
// clamped to the last valid row (their results are never stored).
template <class C>
__device__ __forceinline__ void stage_load(const GemmParams &p, int step, int m0, int n0, int tid,
                                           GemmState<C> &st) {
  static_assert(C::BN % (C::NT / 4) == 0, "operand kind must be uniform per chunk index");
  const bool int4 = step < p.G;
  const uint8_t *wbase = int4 ? p.B4 : p.B8;
  const uint8_t *abase = int4 ? p.A4 : p.A8;
  const int stride = int4 ? p.K4h : kKeeper;
  const int koff = (int4 ? step : step - p.G) * 64;
#pragma unroll
  for (int i = 0; i < C::NCH; ++i) {
    const int row = (tid >> 2) + i * (C::NT / 4);
    const int j = tid & 3;
    const bool isW = i * (C::NT / 4) < C::BN;
    const int idx = isW ? min(n0 + row, p.N - 1) : min(m0 + row - C::BN, p.M - 1);
    const uint8_t *src = (isW ? wbase : abase) + (int64_t)idx * stride + koff + j * 16;
    st.pre[i] = *reinterpret_cast<const v4u *>(src);
  }
  // scales of this step: group scales for int4 steps; the keeper scales ride with the LAST step
  // (step == G stages nothing useful: no dequant happens between the two keeper halves).
  if (tid < C::ROWS) {
    const bool keeper = step > p.G;
    const int g = min(step, p.G - 1);
    if (tid < C::BN) {
      const int n = min(n0 + tid, p.N - 1);
      st.pre_scale = (float)(keeper ? p.sB8[n] : p.sB[(int64_t)g * p.N + n]);
    } else {
      const int m = min(m0 + tid - C::BN, p.M - 1);
      const int off = p.ref_layout ? ref_scale_index(m) : m;
      // int4 operands are staged as 16*value on both sides -> fold 1/256 here (exact power of two)
      st.pre_scale = keeper ? (float)p.sA8[off] : (float)p.sA[(int64_t)g * p.ldA + off] * (1.0f / 256.0f);
    }
  }
}

// ---- registers -> LDS (int4 -> int8*16 widening happens here, once per workgroup) -----------------
template <class C>
__device__ __forceinline__ void stage_store(const GemmParams &p, int step, char *stage, int tid,
                                            const GemmState<C> &st) {
  const bool int4 = step < p.G;
#pragma unroll
  for (int i = 0; i < C::NCH; ++i) {
    const int row = (tid >> 2) + i * (C::NT / 4);
    const int j = tid & 3;
    const int sw = (row >> 1) & 7;
    char *rbase = stage + row * 128;
    const v4u v = st.pre[i];
    if (int4) {
      v4u lo, hi;
      lo = (v << 4) & 0xF0F0F0F0u;     // elements 2i   (low nibbles)  * 16
      hi = v & 0xF0F0F0F0u;            // elements 2i+1 (high nibbles) * 16
      *reinterpret_cast<v4u *>(rbase + (((2 * j) ^ sw) << 4)) = lo;
      *reinterpret_cast<v4u *>(rbase + (((2 * j + 1) ^ sw) << 4)) = hi;
    } else {
      *reinterpret_cast<v4u *>(rbase + ((j ^ sw) << 4)) = v;
    }
  }
  if (tid < C::ROWS) reinterpret_cast<float *>(stage + C::TILE_BYTES)[tid] = st.pre_scale;
}

// ---- one K step out of LDS: fragment reads, MFMAs, fused dequant ----------------------------------
// ABL (tuning only, never the default): 1 = no dequant VALU, 2 = no LDS stage stores, 4 = no MFMA, 8 = no global loads
template <class C, int KSTEPS, bool INIT, bool DEQ, int ABL = 0>
__device__ __forceinline__ void compute_step(const char *stage, int wm, int wn, int lane, v16i (&acc)[C::TN][C::TM],
                                             float (&c)[C::TN][C::TM][16]) {
  const int l31 = lane & 31, h = lane >> 5;
  const char *wt = stage;                       // weights tile, rows = n
  const char *at = stage + C::BN * 128;         // activations tile, rows = m
  v4i af[C::TN][KSTEPS], bf[C::TM][KSTEPS];
#pragma unroll
  for (int tn = 0; tn < C::TN; ++tn) {
    const int row = wn * C::WN + tn * 32 + l31;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
      af[tn][s] = *reinterpret_cast<const v4i *>(wt + row * 128 + (((2 * s + h) ^ sw) << 4));
  }
#pragma unroll
  for (int tm = 0; tm < C::TM; ++tm) {
    const int row = wm * C::WM + tm * 32 + l31;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
      bf[tm][s] = *reinterpret_cast<const v4i *>(at + row * 128 + (((2 * s + h) ^ sw) << 4));
  }
  v16i magic;
#pragma unroll
  for (int i = 0; i < 16; ++i) magic[i] = kMagicBits;

  const float *sBl = reinterpret_cast<const float *>(stage + C::TILE_BYTES);
  const float *sAl = sBl + C::BN;
  float sa[C::TM], nms[C::TM];
  if constexpr (DEQ) {
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
      sa[tm] = sAl[wm * C::WM + tm * 32 + l31];
      nms[tm] = -kMagic * sa[tm];               // exact: 3*2^22 times an 11-bit significand
    }
  }
#pragma unroll
  for (int tn = 0; tn < C::TN; ++tn) {
    v4f sbv[4];
    if constexpr (DEQ) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        sbv[q] = *reinterpret_cast<const v4f *>(sBl + wn * C::WN + tn * 32 + 8 * q + 4 * h);
    }
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        if constexpr (ABL & 4) {
          asm volatile("" ::"v"(af[tn][s]), "v"(bf[tm][s]));
          if (INIT && s == 0) acc[tn][tm] = magic;
        } else if (INIT && s == 0)
          acc[tn][tm] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][s], bf[tm][s], magic, 0, 0, 0);
        else
          acc[tn][tm] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][s], bf[tm][s], acc[tn][tm], 0, 0, 0);
      }
      if constexpr (DEQ && (ABL & 1)) {
        asm volatile("" ::"v"(acc[tn][tm]));
      } else if constexpr (DEQ) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float t = __builtin_fmaf(__int_as_float(acc[tn][tm][r]), sa[tm], nms[tm]);
          c[tn][tm][r] = __builtin_fmaf(t, sbv[r >> 2][r & 3], c[tn][tm][r]);
        }
      }
    }
  }
}

template <class C, int ABL = 0>
__global__ __launch_bounds__(C::NT) void gemm_w4a4_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WGN, wn = wave % C::WGN;

  // block -> tile.  n-tiles fastest so that concurrently running blocks share the activation panel.
  const int nbn = (p.N + C::BN - 1) / C::BN;
  const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
  const int m0 = bm * C::BM, n0 = bn * C::BN;

  v16i acc[C::TN][C::TM];
  float c[C::TN][C::TM][16];
#pragma unroll
  for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[tn][tm][r] = 0.f;

  GemmState<C> st;
  const int nsteps = p.G + 2;       // G int4 groups + 2 keeper halves of 64 int8 columns

  stage_load<C>(p, 0, m0, n0, tid, st);
  stage_store<C>(p, 0, lds, tid, st);
  __syncthreads();

  for (int step = 0; step < nsteps; ++step) {
    char *cur = lds + (step & 1) * C::STAGE_BYTES;
    char *nxt = lds + ((step + 1) & 1) * C::STAGE_BYTES;
    const bool more = step + 1 < nsteps;
    if (more && !(ABL & 8)) stage_load<C>(p, step + 1, m0, n0, tid, st);
    if (step < p.G)
      compute_step<C, 4, true, true, ABL>(cur, wm, wn, lane, acc, c);
    else if (step == p.G)
      compute_step<C, 2, true, false, ABL>(cur, wm, wn, lane, acc, c);
    else
      compute_step<C, 2, false, true, ABL>(cur, wm, wn, lane, acc, c);
    if (more && !(ABL & 2)) stage_store<C>(p, step + 1, nxt, tid, st);
    __syncthreads();
  }

  // epilogue: lane owns token m and features n = base + 8q + 4h + {0..3}: one 8-byte store per q
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int tm = 0; tm < C::TM; ++tm) {
    const int m = m0 + wm * C::WM + tm * 32 + l31;
    if (m >= p.M) continue;
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * C::WN + tn * 32 + 8 * q + 4 * h;
        if (n >= p.N) continue;
        v2u o;
        half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
        for (int k = 0; k < 4; ++k) ov[k] = f2h(c[tn][tm][4 * q + k]);
        *reinterpret_cast<v2u *>(p.D + (int64_t)m * p.N + n) = o;
      }
    }
  }
}

template <class C, int ABL = 0>
static int launch_gemm(const GemmParams &p, hipStream_t s) {
  static bool attr_set = false;     // benign race: idempotent
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_w4a4_kernel<C, ABL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
      return ATOM_ERR_LAUNCH;
    attr_set = true;
  }
  const int nbm = (p.M + C::BM - 1) / C::BM, nbn = (p.N + C::BN - 1) / C::BN;
  hipLaunchKernelGGL((gemm_w4a4_kernel<C, ABL>), dim3((unsigned)(nbm * nbn)), dim3(C::NT), C::LDS_BYTES, s, p);
  return check_launch();
}

}  // namespace atom

using namespace atom;

static int fill_params(GemmParams &p, const void *A4, const void *B4, const void *sA, const void *sB, const void *A8,
                       const void *B8, const void *sA8, const void *sB8, int64_t M, int64_t N, int64_t K_total, int group,
                       int keeper, int scale_layout);

// largest M served by the weight-streaming decode kernel (ATOM_GEMV_MAXM overrides it for tuning)
static int gemv_max_m() {
  static const int v = [] { const char *e = getenv("ATOM_GEMV_MAXM"); return e ? atoi(e) : 7; }();
  return v;
}

extern "C" {

const char *atom_version(void) { return "atom_hip 0.1 (gfx950)"; }

const char *atom_strerror(int code) {
  switch (code) {
    case ATOM_OK: return "ok";
    case ATOM_ERR_INVALID_ARG: return "invalid argument (null pointer or bad enum)";
    case ATOM_ERR_SHAPE: return "unsupported shape / group / keeper";
    case ATOM_ERR_ALIGN: return "pointer not 16-byte aligned";
    case ATOM_ERR_LAUNCH: return "HIP launch failed";
    default: return "unknown error";
  }
}

int atom_gemm_w4a4_f16(const void *A4, const void *B4, const void *sA, const void *sB, const void *A8,
                       const void *B8, const void *sA8, const void *sB8, void *D, int64_t M, int64_t N,
                       int64_t K_total, int group, int keeper, int scale_layout, void *stream) {
  if (!D) return ATOM_ERR_INVALID_ARG;
  GemmParams p;
  const int fst = fill_params(p, A4, B4, sA, sB, A8, B8, sA8, sB8, M, N, K_total, group, keeper, scale_layout);
  if (fst != ATOM_OK) return fst;
  if (!aligned16(D)) return ATOM_ERR_ALIGN;
  p.D = (half_t *)D;
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
  static const int variant = [] { const char *e = getenv("ATOM_GEMM_VARIANT"); return e ? atoi(e) : 0; }();
  if (p.a_wide && variant != 0 && !(variant >= 320 && variant <= 330)) return ATOM_ERR_INVALID_ARG;
  if (p.f6_rows_a && variant != 0) return ATOM_ERR_INVALID_ARG;
  switch (variant) {   // tuning / ablation variants; 0 is the product path
    case 320: case 324: case 325: return launch_gemm_v3(p, variant - 300, hs);
    case 101: return launch_gemm<GemmCfg<128, 256, 2, 4>, 1>(p, hs);
    case 102: return launch_gemm<GemmCfg<128, 256, 2, 4>, 2>(p, hs);
    case 103: return launch_gemm<GemmCfg<128, 256, 2, 4>, 3>(p, hs);
    case 104: return launch_gemm<GemmCfg<128, 256, 2, 4>, 4>(p, hs);
    case 108: return launch_gemm<GemmCfg<128, 256, 2, 4>, 8>(p, hs);
    case 110: return launch_gemm<GemmCfg<128, 256, 2, 4>, 10>(p, hs);
    case 111: return launch_gemm<GemmCfg<128, 256, 2, 4>, 11>(p, hs);
    case 115: return launch_gemm<GemmCfg<128, 256, 2, 4>, 15>(p, hs);
    case 203: return launch_gemm_v2(p, 3, hs);
    case 204: return launch_gemm_v2(p, 4, hs);
    case 1001: case 1002: case 1003: case 1004: case 1008: case 1016: case 1019: case 1023: case 1031: case 1032: case 1033: case 1035: case 1064: case 1128: case 1256:
      return launch_gemm_v2(p, variant, hs);
    case 1: return launch_gemm<GemmCfg<128, 256, 2, 4>>(p, hs);    // v1: register-staged, int8-expanded LDS tiles
    case 2: return launch_gemm_v2(p, 4, hs);
    case 300: case 301: case 302: case 303: case 304: case 305: case 306: return launch_gemm_v3(p, variant - 300, hs);
    case 310: case 311: case 330: {   // traced run: the trace buffer pointer arrives in ATOM_TRACE_PTR (tools/trace_gemm.cpp)
      const char *e = getenv("ATOM_TRACE_PTR");
      if (!e) return ATOM_ERR_INVALID_ARG;
      p.Dsz = reinterpret_cast<half_t *>(strtoull(e, nullptr, 16));
      return launch_gemm_v3(p, variant - 300, hs);
    }
    default:                                                        // product path
      if (p.f6_rows_a) return launch_gemm_f6(p, hs);           // BF6 operands: block-scaled MFMA kernel, 256x256 tiles
      if (p.a_wide) {   // activations pre-widened by the quant kernels: 256x256 tiles once they fill half the chip
        const int64_t cm256 = (M + 255) / 256, cn256 = (N + 255) / 256, t5 = ((M + 63) / 64) * ((N + 127) / 128);
        const int cfg = cm256 * cn256 >= 128 ? 20 : ((t5 > 256 && t5 < 1024) ? 24 : 25);
        return launch_gemm_v3(p, cfg, hs);
      }
      if (M <= gemv_max_m()) {                                      // decode: weight-streaming dot-product kernel
        const int st = launch_gemv(p, hs);
        if (st != ATOM_ERR_SHAPE) return st;
      }
      {   // prefill: LDS-DMA MFMA tile kernel (gemm_w4a4_v3.hip); tile geometry by how many workgroups the shape yields
          // (measured on MI355X, profiles/r01_tile_selection.txt): big tiles only once they fill the chip.
        const int64_t cm256 = (M + 255) / 256, cm64 = (M + 63) / 64, cn256 = (N + 255) / 256, cn128 = (N + 127) / 128;
        int cfg;
        if (cm256 * cn256 >= 1024) cfg = 0;               // 256x256, 8 waves, 1 WG/CU
        else if (cm256 * cn128 >= 512) cfg = 1;           // 256x128, 4 waves, 2 WGs/CU
        else {
          const int64_t t5 = cm64 * cn128;                // 64x128, 2 waves
          cfg = (t5 > 256 && t5 < 1024) ? 4 : 5;          // in between: 64x64, 1 wave, twice the workgroups
        }
        return launch_gemm_v3(p, cfg, hs);
      }
  }
}

static int fill_params(GemmParams &p, const void *A4, const void *B4, const void *sA, const void *sB, const void *A8,
                       const void *B8, const void *sA8, const void *sB8, int64_t M, int64_t N, int64_t K_total, int group,
                       int keeper, int scale_layout) {
  if (!A4 || !B4 || !sA || !sB || !A8 || !B8 || !sA8 || !sB8) return ATOM_ERR_INVALID_ARG;
  const int a_wide = (scale_layout & ATOM_A_WIDE) != 0;
  const int f6 = (scale_layout & ATOM_AB_F6) != 0;
  scale_layout &= ~(ATOM_A_WIDE | ATOM_AB_F6);
  if (a_wide && f6) return ATOM_ERR_INVALID_ARG;
  if (scale_layout != ATOM_SCALE_LAYOUT_REF && scale_layout != ATOM_SCALE_LAYOUT_PLAIN) return ATOM_ERR_INVALID_ARG;
  if (group != kGroup || keeper != kKeeper) return ATOM_ERR_SHAPE;
  if (M < 1 || N < 64 || (N % 64) != 0 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0) return ATOM_ERR_SHAPE;
  if (M > (1 << 24) || N > (1 << 24) || K_total > (1 << 20)) return ATOM_ERR_SHAPE;
  if ((M > N ? M : N) * ((K_total - kKeeper) / 2) >= (int64_t(1) << 32)) return ATOM_ERR_SHAPE;   // 32-bit DMA offsets
  if (a_wide && M * (K_total - kKeeper) >= (int64_t(1) << 32)) return ATOM_ERR_SHAPE;
  if (!aligned16(A4) || !aligned16(B4) || !aligned16(A8) || !aligned16(B8)) return ATOM_ERR_ALIGN;
  if ((reinterpret_cast<uintptr_t>(sB) & 3u) || (reinterpret_cast<uintptr_t>(sB8) & 3u)) return ATOM_ERR_ALIGN;
  p.A4 = (const uint8_t *)A4; p.B4 = (const uint8_t *)B4;
  p.sA = (const half_t *)sA;  p.sB = (const half_t *)sB;
  p.A8 = (const uint8_t *)A8; p.B8 = (const uint8_t *)B8;
  p.sA8 = (const half_t *)sA8; p.sB8 = (const half_t *)sB8;
  p.D = nullptr; p.D4 = nullptr; p.Dsz = nullptr; p.ws = nullptr; p.splits = 1;
  p.M = (int)M; p.N = (int)N;
  p.K4h = (int)((K_total - kKeeper) / 2);
  p.G = (int)((K_total - kKeeper) / kGroup);
  p.ref_layout = scale_layout == ATOM_SCALE_LAYOUT_REF;
  p.a_wide = a_wide;
  p.f6_rows_a = f6 ? (M + 255) / 256 * 256 : 0;              // == atom_f6_rows(): rows per group, padded to the tile
  p.f6_rows_b = f6 ? (N + 255) / 256 * 256 : 0;
  p.ldA = (int64_t)atom_scale_size(M, scale_layout);
  return ATOM_OK;
}

// Split-K policy: shapes that yield fewer than 512 workgroups of the smallest tile are latency-bound (one pass over K per
// workgroup at ~1 us per K-group); split the K loop over up to 8 workgroups and reduce FP32 partials in a second launch.
static int choose_splits(int64_t M, int64_t N, int64_t K_total) {
  if (M <= gemv_max_m()) return 1;                         // decode kernel
  const int64_t tiles = ((M + 63) / 64) * ((N + 127) / 128);
  const int64_t nsteps = (K_total - kKeeper) / kGroup + 2;
  static const int force = [] { const char *e = getenv("ATOM_SPLITS"); return e ? atoi(e) : 0; }();   // tuning only
  if (force > 0) return force > nsteps / 2 ? (int)(nsteps / 2) : force;
  if (tiles >= 512 || nsteps < 8) return 1;
  int64_t s = 1024 / tiles;                                // measured (profiles/r01_gemm_sweeps.txt): 512x4096x4096 1 -> 4
  if (s > 8) s = 8;                                        // splits: 38.4 -> 31.3 us; 256x13824x5120 1 -> 2: 51.3 -> 47.2 us
  if (tiles >= 96 && s > 4) s = 4;
  if (s > nsteps / 4) s = nsteps / 4;
  return s < 2 ? 1 : (int)s;
}

size_t atom_gemm_w4a4_workspace_bytes(int64_t M, int64_t N, int64_t K_total) {
  if (M < 1 || N < 64 || K_total < 256) return 0;
  const int s = choose_splits(M, N, K_total);
  return s > 1 ? (size_t)s * (size_t)M * (size_t)N * sizeof(float) : 0;
}

int atom_gemm_w4a4_f16_ws(const void *A4, const void *B4, const void *sA, const void *sB, const void *A8, const void *B8,
                          const void *sA8, const void *sB8, void *D, int64_t M, int64_t N, int64_t K_total, int group,
                          int keeper, int scale_layout, void *workspace, size_t workspace_bytes, void *stream) {
  const size_t need = (scale_layout & ATOM_AB_F6) ? 0 : atom_gemm_w4a4_workspace_bytes(M, N, K_total);
  if (need == 0 || !workspace || workspace_bytes < need)
    return atom_gemm_w4a4_f16(A4, B4, sA, sB, A8, B8, sA8, sB8, D, M, N, K_total, group, keeper, scale_layout, stream);
  if (!D) return ATOM_ERR_INVALID_ARG;
  GemmParams p;
  const int st = fill_params(p, A4, B4, sA, sB, A8, B8, sA8, sB8, M, N, K_total, group, keeper, scale_layout);
  if (st != ATOM_OK) return st;
  if (!aligned16(D) || !aligned16(workspace) || (N % 8) != 0) return ATOM_ERR_ALIGN;
  p.D = (half_t *)D;
  p.ws = (float *)workspace;
  p.splits = choose_splits(M, N, K_total);
  return launch_gemm_v3(p, p.a_wide ? 25 : 5, reinterpret_cast<hipStream_t>(stream));
}

int atom_gemm_w4a4_o4(const void *A4, const void *B4, const void *sA, const void *sB, const void *A8, const void *B8,
                      const void *sA8, const void *sB8, void *D_u4, void *D_scale_zero, int64_t M, int64_t N,
                      int64_t K_total, int group, int keeper, int scale_layout, void *stream) {
  if (!D_u4 || !D_scale_zero) return ATOM_ERR_INVALID_ARG;
  GemmParams p;
  const int st = fill_params(p, A4, B4, sA, sB, A8, B8, sA8, sB8, M, N, K_total, group, keeper, scale_layout);
  if (st != ATOM_OK) return st;
  if ((N % 128) != 0) return ATOM_ERR_SHAPE;
  if (p.a_wide || p.f6_rows_a) return ATOM_ERR_INVALID_ARG;  // the u4 epilogue kernel takes packed activations only
  if (!aligned16(D_u4)) return ATOM_ERR_ALIGN;
  p.D4 = (uint8_t *)D_u4;
  p.Dsz = (half_t *)D_scale_zero;
  return launch_gemm_v2_o4(p, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
