// W4A4 GEMM, prefill kernel on the block-scaled MFMA of gfx950 with BF6 operands ("F6" native format).
//
// gfx950 has no INT4 MFMA; the INT8 path (gemm_w4a4_v3.hip) spends 4 x v_mfma_i32_32x32x32_i8 per 128-channel group and
// 4.5 VALU per MFMA widening nibbles.  But every INT4 code in [-8, 7] IS a BF6 (E3M2) number, products of two codes and
// sums of 128 of them are exact in FP32, and v_mfma_scale_f32_32x32x64_f8f6f4 runs BF6 at the FP4 rate: with unit block
// scales (E8M0 127) it is an exact integer dot product of 64 channels at 21.7 ns per instruction where the INT8
// 32x32x32 takes 17.6 ns for 32 channels (tools/probes/bf6_probe.cpp: 0 mismatches, 6.19 vs 3.81 Pop/s) -- 38 % less
// matrix time per group, and no widening at all when both operands arrive as BF6.
//
// Operand format "F6" (produced by the activation quantisers with ATOM_QUANT_F6_CODES and by atom_repack_weight_f6):
//   [G][rows_pad][104 bytes]   group-major; 96 bytes = the group's 128 codes as a little-endian stream of 6-bit BF6
//                              fields; activations: byte 96..97 = the fp16 scale of (row, group), 100..103 = the same as
//                              float32 (weights: zero; their scales are a dense fp16 array, optionally repeated as float32
//                              [G][rows_pad] behind the records: ATOM_B_F6S)
// Group-major makes the 256 rows a tile needs for one K step ONE contiguous 26 KiB block: the LDS-DMA is base + 16*lane
// with no per-row address, rows_pad (a multiple of 256) keeps tail tiles in bounds, and the 104-byte pitch (26 dwords =
// 2 x odd) makes the 32 rows of an MFMA fragment read (3 x ds_read_b64 of 24 bytes) hit 32 distinct bank pairs with no
// swizzle.  The token scale is read from the row itself (no scale DMA, no scale region); weight scales keep their dense
// fp16 array.  The INT8 keeper runs as the two 64-column half-steps of the INT8 kernel in the same stage buffer.
// Arithmetic is the same contract: s = sA * sB (exact), c = fma(idot, s, c) per group in order, keeper last --
// results are bit-identical to the INT8 kernels, except where two or four groups of waves share a tile (KG = 2 / 4, shapes of
// at most 256 tiles): those sum consecutive ranges of the K steps and add the partial sums in order.
//
// Kernels in this file, by generation: gemm_w4a4_f6_kernel (32x32x64 MFMA; tuning and the 64x128 split-K geometry),
// gemm_w4a4_f6x16_kernel (16x16x128 micro-tiles; the 128x128 / 64x128 geometries incl. the K-group variants),
// gemm_w4a4_f6p_kernel (256x256, pipelined across K steps, fp16 weight scales), gemm_w4a4_f6q_kernel (256x256, the headline;
// also the fused gate/up + SiLU x up + quantiser epilogue).  launch_gemm_f6() at the end maps the cfg numbers.
#include <type_traits>
#include "common.h"
#include "quant_math.h"

namespace atom {

__global__ void splitk_reduce_kernel(const float *ws, half_t *D, int64_t MN, int splits);   // gemm_w4a4_v3.hip

namespace f6 {

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int TN = 2;                                    // a wave covers 32*TM tokens x 64 features
constexpr int PITCH = 104;                               // bytes per row and group
constexpr float kMagic = 12582912.0f;
constexpr int kMagicBits = 0x4B400000;

// Tile geometry.  Three instances: 256x256 (8 waves, one workgroup per CU) once the shape yields ~200 of them,
// 256x128 (4 waves, two workgroups per CU, two stages), and 64x128 (2 waves, several workgroups per CU, split-K) for
// skinny M.
template <int BM_, int BN_, int TM_, int NS_, int OCC_ = 2, int FS_ = 0>
struct Cfg {
  // FS = 1: float32 weight scales staged per int4 step (ATOM_B_F6S); FS = 2: fp16 weight scales staged (as FS = 0: 256 B less per
  // stage -- three workgroups of the 128x128 geometry fit a CU only with those) but converted once per step; both: float32 token
  // scale from the record, unscaled MFMA
  static constexpr int FS = FS_;
  static constexpr int FS32 = FS_ == 1;
  static constexpr int OCC = OCC_;                                     // waves per SIMD the register budget is set for
  static constexpr int BM = BM_, BN = BN_, TM = TM_, NS = NS_;
  static constexpr int WM = 32 * TM, WGM = BM / WM, WGN = BN / 64, NW = WGM * WGN, NT = NW * 64;
  static constexpr int W_BYTES = BN * PITCH, A_BYTES = BM * PITCH;
  static constexpr int NBW = W_BYTES / 1024;                           // whole 1 KiB DMA blocks of the weight rows
  static constexpr int NBA = (A_BYTES + 1023) / 1024;                  // activation rows: the last block may be partial
  static constexpr int A_TAIL = (A_BYTES % 1024) / 16;                 // lanes of that partial block (0 = it is whole)
  static constexpr int NSB8 = BN / 128;                                // weight-scale pieces (128 fp16 = 64 dwords each)
  static constexpr int NSB = FS32 ? BN / 64 : NSB8;                    // ... of an int4 step (float32: 64 per piece)
  static constexpr int A_OFF = W_BYTES;
  // BN weight scales (int4 steps: fp16 or float32; keeper: fp16): behind the activation blocks, or -- when the last, partial block
  // leaves room (64-token tiles: 512 B) -- in its unwritten tail, which makes the 64x128 stage exactly 20 KiB
  static constexpr bool SB_IN_TAIL = A_TAIL != 0 && NBA * 1024 - A_BYTES >= BN * (FS32 ? 4 : 2);
  static constexpr int SB_OFF = SB_IN_TAIL ? W_BYTES + A_BYTES : W_BYTES + NBA * 1024;
  static constexpr int STAGE_BYTES = SB_IN_TAIL ? W_BYTES + NBA * 1024 : SB_OFF + BN * (FS32 ? 4 : 2);
  static constexpr int KP_SA_OFF = (BN + BM) * 64;                     // keeper half-steps: rows of 64 B, then BM dwords sA8
  static constexpr int NPIECE = NBW + NBA + NSB;                       // DMA instructions per int4 stage
  static constexpr int NKP = (BN + BM) / 16 + BM / 64 + NSB8;          // ... per keeper half-step
  // The LDS-DMA instructions are issued by the first NDW waves.
  static constexpr int NDW = NW;   // (measured: NW / 2 -- only the older wave of each SIMD issuing -- is 6 % slower)
  static constexpr int GLDS = ((NPIECE > NKP ? NPIECE : NKP) + NDW - 1) / NDW;  // per issuing wave, padded with repeats
  static constexpr int EP_BYTES = NW * (WM < 64 ? WM : 64) * 144;
  static constexpr int LDS_BYTES = NS * STAGE_BYTES > EP_BYTES ? NS * STAGE_BYTES : EP_BYTES;
  static_assert(BN % 128 == 0 && BM % 64 == 0 && BM % WM == 0 && (TM == 1 || TM == 2 || TM == 4) && NS >= 2, "geometry");
  static_assert(KP_SA_OFF + BM * 4 <= SB_OFF && LDS_BYTES <= 160 * 1024, "stage layout");
};

// DMA instruction i (of C::GLDS) of this wave for int4 group g
template <class C>
__device__ __forceinline__ void issue_int4_piece(const GemmParams &p, int g, char *slot, int wave, int lane, int m0, int n0, int i) {
  const uint8_t *wsrc = p.B4 + ((int64_t)g * p.f6_rows_b + n0) * PITCH;
  const uint8_t *asrc = p.A4 + ((int64_t)g * p.f6_rows_a + m0) * PITCH - C::W_BYTES;   // block j >= NBW is asrc + j*1024
  const unsigned sl = lds_addr(slot);                      // (one cast per stage; integer offsets per piece)
  int j = i * C::NDW + wave;                               // piece j: NBW weight blocks, NBA activation blocks, NSB scales
  j = j < C::NPIECE ? j : C::NPIECE - C::NSB + (j - C::NPIECE) % C::NSB;   // padding repeats a scale piece (same bytes, same place)
  if (i * C::NDW + C::NDW <= C::NBW + C::NBA - (C::A_TAIL ? 1 : 0)) {              // compile time: whole data blocks only
    const uint8_t *base = (i * C::NDW + C::NDW <= C::NBW || j < C::NBW) ? wsrc : asrc;
    lds_dma_sv<16>(base + j * 1024, (unsigned)lane * 16u, sl + j * 1024);
  } else if (j < C::NBW + C::NBA) {
    const uint8_t *base = j < C::NBW ? wsrc : asrc;
    // the partial last activation block runs with fewer lanes enabled (one instruction either way: vmcnt stays uniform)
    if (!C::A_TAIL || j < C::NBW + C::NBA - 1 || lane < C::A_TAIL)
      lds_dma_sv<16>(base + j * 1024, (unsigned)lane * 16u, sl + j * 1024);
  } else if constexpr (C::FS32) {                          // 64 float32 weight scales per piece (rows padded: no clamp)
    const int part = j - (C::NBW + C::NBA);
    lds_dma_sv<4>(p.sB32 + (int64_t)g * p.f6_rows_b + n0 + part * 64, (unsigned)lane * 4u, sl + C::SB_OFF + part * 256);
  } else {                                                 // 128 weight scales per piece, a dword (2 channels) per lane
    const int part = j - (C::NBW + C::NBA);
    const half_t *sBb = p.sB + (int64_t)g * p.N;
    const int n = min(n0 + part * 128 + 2 * lane, p.N - 2);
    lds_dma_at<4>(sBb + n, sl + C::SB_OFF + part * 256);
  }
}

template <class C>
__device__ __forceinline__ void issue_int4(const GemmParams &p, int g, char *slot, int wave, int lane, int m0, int n0) {
#pragma unroll
  for (int i = 0; i < C::GLDS; ++i) issue_int4_piece<C>(p, g, slot, wave, lane, m0, n0, i);
}

// keeper half-step `half` (0 / 1): the INT8 kernel's layout -- 16 rows x 64 B per DMA block, XOR-swizzled chunks
template <class C>
__device__ __forceinline__ void issue_keeper(const GemmParams &p, int half, char *slot, int wave, int lane, int m0, int n0) {
  // only two of these per tile: addresses are computed here instead of living in registers through the int4 loop
  const unsigned kj = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  constexpr int ND = (C::BN + C::BM) / 16, NSA = C::BM / 64;
  const unsigned sl = lds_addr(slot);
#pragma unroll
  for (int i = 0; i < C::GLDS; ++i) {
    int j = i * C::NDW + wave;
    j = j < C::NKP ? j : ND + NSA + (j % C::NSB8);         // padding repeats a weight-scale piece
    if (i * C::NDW + C::NDW <= ND || j < ND) {               // (first half: known at compile time)
      const int row = j * 16 + (lane >> 2);
      const unsigned idx = (unsigned)(j < C::BN / 16 ? min(n0 + row, p.N - 1) : min(m0 + row - C::BN, p.M - 1));
      const uint8_t *base = (j < C::BN / 16 ? p.B8 : p.A8) + half * 64;
      lds_dma_at<16>(base + idx * kKeeper + kj, sl + j * 1024);
    } else if (j < ND + NSA) {                             // sA8 of 64 tokens: one fp16 per lane -> zero-extended dword
      const int pc = j - ND;
      const int idx = min(m0 + pc * 64 + lane, p.M - 1);
      const unsigned ksa = (unsigned)(p.ref_layout ? ref_scale_index(idx) : idx);
      lds_dma_at<2>(p.sA8 + ksa, sl + C::KP_SA_OFF + pc * 256);
    } else {
      const int part = j - ND - NSA;
      const int n = min(n0 + part * 128 + 2 * lane, p.N - 2);
      lds_dma_at<4>(p.sB8 + n, sl + C::SB_OFF + part * 256);
    }
  }
}

__device__ __forceinline__ v8i frag24(const char *p) {    // 24 bytes = 32 BF6 fields -> MFMA operand (6 of 8 VGPRs)
  const v2u a = *reinterpret_cast<const v2u *>(p);
  const v2u b = *reinterpret_cast<const v2u *>(p + 8);
  const v2u c = *reinterpret_cast<const v2u *>(p + 16);
  return v8i{(int)a.x, (int)a.y, (int)b.x, (int)b.y, (int)c.x, (int)c.y, 0, 0};
}

__device__ __forceinline__ void dequant16(float (&acc)[16], float sa, const char *psb, float (&c)[16]) {
  // (the 16 weight scales are re-read per tile: holding a step's 32 in registers puts them into the LDS burst behind the
  // barrier and measures 3 % slower)
  v2u sbp[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) sbp[q] = *reinterpret_cast<const v2u *>(psb + 16 * q);
  // the contract (round 5): s = sA * sB -- exact in FP32 --, c = fma(idot, s, c).  All 16 products FIRST, then the FMAs: a multiply
  // followed directly by the FMA that reads it costs 3.2 cycles per instruction instead of 2 (dependent issue)
  float sc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const half_t *hv = reinterpret_cast<const half_t *>(&sbp[r >> 2]);
    sc[r] = (float)hv[r & 3] * sa;
    asm volatile("" : "+v"(sc[r]));
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    c[r] = __builtin_fmaf(acc[r], sc[r], c[r]);
    asm volatile("" : "+v"(c[r]));
  }
}

// one int4 group out of LDS: 2 BF6 MFMAs per 32x32 tile
// ABL (tools only, -DATOM_TOOLS): 2 = no de-quantisation, 4 = no MFMA, 8 = no fragment refills, 16 = s_memtime
// stamps, 32 = no priority swap
struct NoDma { __device__ __forceinline__ void operator()(int) const {} };

// `dma(i)`, i < C::GLDS: the wave's LDS-DMA instructions for a later stage, spread over the tiles -- each is issued behind a
// tile's MFMA pair, where the wave would wait for the matrix pipe anyway (issuing them in a block at the top of the step
// stalls both waves of a SIMD on the address path at the same time)
template <class C, int ABL = 0, class F = NoDma>
__device__ __forceinline__ void compute_int4(const char *slot, int wm, int wn, int lane, float (&c)[TN][C::TM][16], F dma = F(),
                                             unsigned long long *tp = nullptr, bool older = false) {
  constexpr int TM = C::TM;
  const int l31 = lane & 31, h = lane >> 5;
  const char *pw = slot + (wn * 64 + l31) * PITCH + h * 24;                   // + tn*32*PITCH + s*48
  const char *pa = slot + C::A_OFF + (wm * C::WM + l31) * PITCH + h * 24;     // + tm*32*PITCH + s*48
  const char *psb = slot + C::SB_OFF + (wn * 64 + 4 * h) * 2;                 // + tn*64 + 16*q
  // only what the first tile needs is loaded ahead of its MFMAs (all 8 waves hit the LDS at once behind the barrier); the
  // second feature fragment pair follows behind the first MFMA pair
  v8i af[TN][2];
  constexpr bool DB = C::OCC <= 2;                 // token fragments double-buffered across tm when the register budget allows
  constexpr int NB = DB ? 2 : 1;
  v8i bf[NB][2];
  af[0][0] = frag24(pw);
  bf[0][0] = frag24(pa);
  af[0][1] = frag24(pw + 48);
  bf[0][1] = frag24(pa + 48);
  half_t sah = *reinterpret_cast<const half_t *>(pa - h * 24 + 96);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    __builtin_amdgcn_sched_barrier(0);
    const float sa = (float)sah;
    if (tm + 1 < TM) sah = *reinterpret_cast<const half_t *>(pa - h * 24 + 96 + (tm + 1) * 32 * PITCH);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      __builtin_amdgcn_sched_barrier(0);
      v16f acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      if constexpr (C::NW >= 8 && TM == 4 && !(ABL & 32)) {
        // two waves of the workgroup per SIMD: they swap priority mid-step.  The arbiter otherwise always serves the older
        // wave first; it then idles ~1k cycles at every barrier while the younger one finishes the step alone at the
        // single-wave issue rate (s_memtime trace: barrier wait 1050 -> 400 cycles, step 4750 -> 4500; every-two-tiles
        // swapping measures slower than no swapping)
        if (tm * TN + tn == 0) { if (older) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2); }
        if (tm * TN + tn == TM * TN / 2) { if (older) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
      }
      if constexpr ((ABL & 16) != 0)
        if (tp && tm == 0 && tn == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tp[3] = __builtin_amdgcn_s_memtime(); }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        // cbsz = blgp = 3: BF6 (E3M2) x BF6; block scales E8M0 127 = 2^0
        if constexpr (!(ABL & 4)) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[tn][s], bf[tm % NB][s], acc, 3, 3, 0, 127, 0, 127);
        else asm volatile("" : "+v"(acc) : "v"(af[tn][s]), "v"(bf[tm % NB][s]));
      }
      if (tn == (DB ? 0 : TN - 1) && tm + 1 < TM) {       // next token fragments: behind this tm's first MFMA pair, or (single
        __builtin_amdgcn_sched_barrier(0);                // buffer) in place behind its last one
        if (!(ABL & 8)) {
          bf[(tm + 1) % NB][0] = frag24(pa + (tm + 1) * 32 * PITCH);
          bf[(tm + 1) % NB][1] = frag24(pa + (tm + 1) * 32 * PITCH + 48);
        } else if (DB) {
          bf[(tm + 1) % NB][0] = bf[tm % NB][0];
          bf[(tm + 1) % NB][1] = bf[tm % NB][1];
        }
      }
      if (tm == 0 && tn == 0) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t2 = 1; t2 < TN; ++t2)
#pragma unroll
          for (int s = 0; s < 2; ++s) af[t2][s] = frag24(pw + t2 * 32 * PITCH + s * 48);
      }
      {
        constexpr int NTILE = TM * TN;
        const int t = tm * TN + tn;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = (t * C::GLDS + NTILE - 1) / NTILE; i < ((t + 1) * C::GLDS + NTILE - 1) / NTILE; ++i) dma(i);   // front-loaded
        __builtin_amdgcn_sched_barrier(0);
      }
      float a16[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) a16[i] = acc[i];
      if constexpr (!(ABL & 2)) dequant16(a16, sa, psb + tn * 64, c[tn][tm]);
      else { c[tn][tm][0] += a16[0] + a16[15]; asm volatile("" ::"v"(acc)); }
      if constexpr ((ABL & 16) != 0) {                     // tools only: s_memtime after tiles 1, 3, 5, 7
        const int t = tm * TN + tn;
        if (tp && (t & 1)) { __builtin_amdgcn_sched_barrier(0); tp[4 + (t >> 1)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
      }
    }
  }
}

// The keeper out of LDS (INT8 MFMA, magic-biased accumulator): its two 64-column halves sit in two stage slots and are multiplied
// in ONE step -- four chained MFMAs per 32x32 tile into one accumulator, one de-quantisation (the contract of include/atom_hip.h; the
// reference kernel accumulates both keeper k-steps before its dequant too, Dense_layer_gemm_i4_o16.cuh:640-691)
template <class C>
__device__ __forceinline__ void compute_keeper(const char *slot, const char *slot1, int wm, int wn, int lane, float (&c)[TN][C::TM][16]) {
  constexpr int TM = C::TM;
  const int l31 = lane & 31, h = lane >> 5;
  const int sw = (l31 >> 2) & 3;
  const int ow0 = (wn * 64 + l31) * 64 + (((0 + h) ^ sw) << 4), ow1 = (wn * 64 + l31) * 64 + (((2 + h) ^ sw) << 4);
  const int oa0 = (C::BN + wm * C::WM + l31) * 64 + (((0 + h) ^ sw) << 4), oa1 = (C::BN + wm * C::WM + l31) * 64 + (((2 + h) ^ sw) << 4);
  const char *psa = slot + C::KP_SA_OFF + (wm * C::WM + l31) * 4;
  const char *psb = slot + C::SB_OFF + (wn * 64 + 4 * h) * 2;
  v4i af[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    af[tn][0] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot + ow0 + tn * 2048));
    af[tn][1] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot + ow1 + tn * 2048));
    af[tn][2] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot1 + ow0 + tn * 2048));
    af[tn][3] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot1 + ow1 + tn * 2048));
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const v4i b0 = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot + oa0 + tm * 2048));
    const v4i b1 = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot + oa1 + tm * 2048));
    const v4i b2 = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot1 + oa0 + tm * 2048));
    const v4i b3 = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot1 + oa1 + tm * 2048));
    const float sa = (float)*reinterpret_cast<const half_t *>(psa + tm * 128);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      __builtin_amdgcn_sched_barrier(0);
      v16i magic;
#pragma unroll
      for (int i = 0; i < 16; ++i) magic[i] = kMagicBits;
      v16i a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][0], b0, magic, 0, 0, 0);
      a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][1], b1, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][2], b2, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][3], b3, a, 0, 0, 0);
      v2u sbp[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) sbp[q] = *reinterpret_cast<const v2u *>(psb + (tn * 32 + 8 * q) * 2);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const half_t *hv = reinterpret_cast<const half_t *>(&sbp[r >> 2]);
        const float idot = __int_as_float(a[r]) - kMagic;               // exact: the register read as a float is 12582912 + idot
        c[tn][tm][r] = __builtin_fmaf(idot, (float)hv[r & 3] * sa, c[tn][tm][r]);
        asm volatile("" : "+v"(c[tn][tm][r]));
      }
    }
  }
}

template <class C, bool SK, int ABL = 0>
__global__ __launch_bounds__(C::NT, C::OCC) void gemm_w4a4_f6_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int TM = C::TM, NS = C::NS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < C::NW);
  const int wm = wave / C::WGN, wn = wave % C::WGN;

  const int nbn = (p.N + C::BN - 1) / C::BN, nbm = (p.M + C::BM - 1) / C::BM;
  const int nwg = nbm * nbn;
  int id = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  constexpr int GM = 4;
  const int band = id / (GM * nbn), inband = id % (GM * nbn);
  const int rows_in_band = min(GM, nbm - band * GM);
  const int bm = band * GM + inband % rows_in_band, bn = inband / rows_in_band;
  const int m0 = bm * C::BM, n0 = bn * C::BN;

  float c[TN][TM][16];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[a][b][r] = 0.f;

  // K steps: the G int4 groups, then the keeper = G + 1 compute steps; DMA stages: G + 2 (the keeper's two 64-column halves).
  // split-K (SK): blockIdx.y owns the compute steps [cb, ce) and writes FP32 partial sums to p.ws
  const int csteps = p.G + 1;
  const int s_begin = SK ? (int)((int64_t)csteps * blockIdx.y / p.splits) : 0;
  const int c_end = SK ? (int)((int64_t)csteps * (blockIdx.y + 1) / p.splits) : csteps;
  const bool has_keeper = c_end == csteps;
  const int nsteps = c_end + (has_keeper ? 1 : 0);          // DMA stages [s_begin, nsteps)
  auto issue = [&](int step) {
    char *slot = lds + (step % NS) * C::STAGE_BYTES;
    const int s = min(step, nsteps - 1);
    if (wave >= C::NDW) return;
    if (s < p.G) issue_int4<C>(p, s, slot, wave, lane, m0, n0);
    else issue_keeper<C>(p, s - p.G, slot, wave, lane, m0, n0);
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s_begin + s);
  unsigned long long *tbase = nullptr;                   // tools only (ABL & 16): u64 [wave][64 steps][8] stamps of workgroup 0
  if constexpr ((ABL & 16) != 0)
    if (blockIdx.x == 0 && lane == 0) tbase = reinterpret_cast<unsigned long long *>(p.Dsz) + wave * 64 * 8;
#define ATOM_F6_STAMP(k) if constexpr ((ABL & 16) != 0) { if (tbase) tbase[step * 8 + k] = __builtin_amdgcn_s_memtime(); }
#define ATOM_F6_STEP(ISSUE, COMPUTE)                                                     \
  {                                                                                      \
    ATOM_F6_STAMP(0)                                                                     \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ABL & 1) ? 0 : C::GLDS * (NS - 2)) : "memory"); \
    ATOM_F6_STAMP(1)                                                                     \
    __builtin_amdgcn_s_barrier();                                                        \
    ATOM_F6_STAMP(2)                                                                     \
    ISSUE;                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    COMPUTE;                                                                             \
  }
  int step = s_begin;
  // int4 steps whose prefetch (step + NS - 1) is an int4 group too: DMA instructions interleaved with the tiles
  for (; step + NS - 1 < min(p.G, nsteps); ++step) {
    char *nslot = lds + ((step + NS - 1) % NS) * C::STAGE_BYTES;
    const int g = step + NS - 1;
    auto dma = [&](int i) { if (!(ABL & 1) && wave < C::NDW) issue_int4_piece<C>(p, g, nslot, wave, lane, m0, n0, i); };
    ATOM_F6_STEP((void)0, (compute_int4<C, ABL>(lds + (step % NS) * C::STAGE_BYTES, wm, wn, lane, c, dma, tbase ? tbase + step * 8 : nullptr, wave < C::NW / 2)))
  }
  for (; step < min(p.G, nsteps); ++step)
    ATOM_F6_STEP(if (!(ABL & 1)) issue(step + NS - 1), (compute_int4<C, ABL>(lds + (step % NS) * C::STAGE_BYTES, wm, wn, lane, c, NoDma(), nullptr, wave < C::NW / 2)))
  __builtin_amdgcn_s_setprio(0);
#undef ATOM_F6_STEP
  if (has_keeper) {                                        // step == G: both halves (stages G, G + 1) in one compute step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (NS == 2) {                               // (two-stage ring: the second half only fits once stage G - 1 is read)
      if (wave < C::NDW) issue_keeper<C>(p, 1, lds + ((p.G + 1) % NS) * C::STAGE_BYTES, wave, lane, m0, n0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    compute_keeper<C>(lds + (p.G % NS) * C::STAGE_BYTES, lds + ((p.G + 1) % NS) * C::STAGE_BYTES, wm, wn, lane, c);
    ++step;
  }
  { ATOM_F6_STAMP(0) }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int l31 = lane & 31, h = lane >> 5;
  if constexpr (SK) {
    // FP32 partial tile: lane owns token m and 4 consecutive features per (tile, q) -> one 16-byte store each
    float *wsp = p.ws + (int64_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = m0 + wm * C::WM + tm * 32 + l31;
      if (m >= p.M) continue;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 64 + tn * 32 + 8 * q + 4 * h;
          if (n >= p.N) continue;
          *reinterpret_cast<v4f *>(wsp + (int64_t)m * p.N + n) =
              v4f{c[tn][tm][4 * q], c[tn][tm][4 * q + 1], c[tn][tm][4 * q + 2], c[tn][tm][4 * q + 3]};
        }
    }
    return;
  }
  constexpr int EP_STRIDE = 144;
  char *ep = lds + wave * (64 * EP_STRIDE);
#pragma unroll
  for (int half = 0; half < TM / 2; ++half) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const int tm = half * 2 + t2;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v2u o;
          half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
          for (int k = 0; k < 4; ++k) ov[k] = f2h(c[tn][tm][4 * q + k]);
          *reinterpret_cast<v2u *>(ep + (t2 * 32 + l31) * EP_STRIDE + (tn * 32 + 8 * q + 4 * h) * 2) = o;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rl = i * 8 + (lane >> 3);
      const int ch = lane & 7;
      const v4u v = *reinterpret_cast<const v4u *>(ep + rl * EP_STRIDE + ch * 16);
      const int m = m0 + wm * C::WM + half * 64 + rl;
      const int n = n0 + wn * 64 + ch * 8;
      if (m < p.M && n < p.N) *reinterpret_cast<v4u *>(p.D + (int64_t)m * p.N + n) = v;
    }
  }
}

// ================================================================================================================
// The 256x256 and 128x128 geometries on v_mfma_scale_f32_16x16x128_f8f6f4 (their product kernel): one MFMA covers a whole
// 128-channel group of a 16x16 tile -- no dependent MFMA pair -- and its result is 4 VGPRs, so a second accumulator set
// costs 8 registers instead of 16 and the next tiles' MFMAs are issued ahead of the previous tiles' de-quantisation.
// Measured against the 32x32x64 kernel above (same contract, bit-identical output): 67.4 vs 69.7 us at 4096^3, 485 vs 501 us
// at 8192^3; tiles in pairs (2 MFMAs in flight, then 8 multiplies + 8 FMAs) beat single tiles and quads (69.1 us each), one
// pair of look-ahead beats two (68.7), and the mid-step priority swap of the two waves of a SIMD is worth 6 % here (71.2
// without).  128x128 (three workgroups per CU): 43.1 vs 44.1 us at 2048x4096x4096, 107 vs 110 us at 2048x11008x4096.
// Wave tile: 64 features (4 blocks fb) x 128 tokens (8 blocks tb).  Lane l: MFMA row / column l % 16, k-block l / 16
// (32 codes = 24 bytes at byte 24 * (l / 16) of the row); result: token l % 16, features 4 * (l / 16) + r.
typedef float v4f_t __attribute__((ext_vector_type(4)));

// De-quantisation of a PAIR of micro-tiles (accumulators acc[0], acc[1]: exact integer dot products as floats) for one token scale.
// The contract (include/atom_hip.h, round 5): s = sA * sB -- exact in FP32, both are fp16 values --, c = fma(idot, s, c): the
// reference's dequant shape (Dense_layer_gemm_i4_o16.cuh:413-431: scale product first, then accu += c_frag * rs_scale).
//   INTER = true  (fragment rows interleaved: row i of micro-tile k = feature 2 i + k): sb[2 r + k] is the scale of (k, r);
//   INTER = false (plain rows: micro-tile k holds 16 consecutive features): sb[4 k + r];
//   PAIR  = the caller asserts that output channels 2 j, 2 j + 1 share their weight scale (weight_channel_group = 2, ATOM_B_SCALE_PAIRS:
//           what the reference kernel requires -- it applies ONE product to both columns of a pair, :419-431): with interleaved rows
//           the two micro-tiles then share all four products -- 4 multiplies + 8 FMAs instead of 8 + 8 -- and sb holds the four
//           distinct scales only (sb[r] = channels 2 r, 2 r + 1 of the lane's eight).
// All products first, then the FMAs: a multiply followed directly by the FMA that reads it issues at 3.2 cycles instead of 2.
template <bool PAIR, bool INTER = true>
__device__ __forceinline__ void deq_pair(const v4f_t &a0, const v4f_t &a1, float sa, const float *sb, float (&c0)[4], float (&c1)[4]) {
  if constexpr (PAIR) {
    static_assert(INTER, "shared products need the interleaved rows");
    float s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = sa * sb[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) { c0[r] = __builtin_fmaf(a0[r], s[r], c0[r]); asm volatile("" : "+v"(c0[r])); }
#pragma unroll
    for (int r = 0; r < 4; ++r) { c1[r] = __builtin_fmaf(a1[r], s[r], c1[r]); asm volatile("" : "+v"(c1[r])); }
  } else if constexpr (INTER) {
    float s[8];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[4 * k + r] = sa * sb[2 * r + k];
#pragma unroll
    for (int r = 0; r < 4; ++r) { c0[r] = __builtin_fmaf(a0[r], s[r], c0[r]); asm volatile("" : "+v"(c0[r])); }
#pragma unroll
    for (int r = 0; r < 4; ++r) { c1[r] = __builtin_fmaf(a1[r], s[4 + r], c1[r]); asm volatile("" : "+v"(c1[r])); }
  } else {                                                 // (128-register geometries: one micro-tile's four products at a time)
    float s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = sa * sb[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) { c0[r] = __builtin_fmaf(a0[r], s[r], c0[r]); asm volatile("" : "+v"(c0[r])); }
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = sa * sb[4 + r];
#pragma unroll
    for (int r = 0; r < 4; ++r) { c1[r] = __builtin_fmaf(a1[r], s[r], c1[r]); asm volatile("" : "+v"(c1[r])); }
  }
}

__device__ __forceinline__ void dequant4x(const v4f_t &acc, float sa, const v2u &sb, float (&c)[4]) {
  const half_t *hv = reinterpret_cast<const half_t *>(&sb);
  float t[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = (float)hv[r] * sa;                    // the exact scale products
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    c[r] = __builtin_fmaf(acc[r], t[r], c[r]);
    asm volatile("" : "+v"(c[r]));
  }
}

// PH: a workgroup barrier in the middle of the step (the two K groups of a workgroup run half a step apart: one group's
// fragment loads behind its step-start barrier meet the other group's second half of MFMAs instead of its loads)
template <class C, class F = NoDma, bool PRIO = (C::NW >= 8), bool PH = false>
__device__ __forceinline__ void compute_int4_x16(const char *slot, int wm, int wn, int lane, float (&c)[4][C::WM / 16][4], F dma = F(),
                                                 bool older = false, unsigned *tp = nullptr) {
  const int l15 = lane & 15, kb = lane >> 4;
  const char *pw = slot + (wn * 64 + l15) * PITCH + kb * 24;                  // + fb*16*PITCH
  constexpr int NTB = C::WM / 16;                                             // token blocks of the wave tile
  const char *pa = slot + C::A_OFF + (wm * C::WM + l15) * PITCH + kb * 24;    // + tb*16*PITCH
  const char *psa = slot + C::A_OFF + (wm * C::WM + l15) * PITCH + 96;        // + tb*16*PITCH
  const char *psb = slot + C::SB_OFF + (wn * 64 + 4 * kb) * (C::FS32 ? 4 : 2);   // + fb*32 (float32: fb*64)
  v8i af[4], bf[2];
  // scales: float32 straight into the de-quantisation (C::FS: the record's float32 token scale; the weight scales staged as float32
  // (FS = 1) or converted once per step (FS = 2)) or fp16 converted per use (FS = 0)
  using sa_t = typename std::conditional<C::FS != 0, float, half_t>::type;
  using sb_t = typename std::conditional<C::FS != 0, v4f_t, v2u>::type;
  constexpr int SA_AT = C::FS ? 4 : 0;
  sb_t sb[4];
  // fragment loads in the order the MFMAs consume them (all 8 waves hit the LDS at once behind the barrier)
  bf[0] = frag24(pa);
  af[0] = frag24(pw);
  sa_t sah = *reinterpret_cast<const sa_t *>(psa + SA_AT);
#pragma unroll
  for (int fb = 1; fb < 4; ++fb) af[fb] = frag24(pw + fb * 16 * PITCH);
#pragma unroll
  for (int fb = 0; fb < 4; ++fb) {
    if constexpr (C::FS == 2) {
      const v2u raw = *reinterpret_cast<const v2u *>(psb + fb * 32);
      const half_t *hv = reinterpret_cast<const half_t *>(&raw);
      sb[fb] = v4f_t{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
    } else {
      sb[fb] = *reinterpret_cast<const sb_t *>(psb + fb * (C::FS32 ? 64 : 32));
    }
  }
  // tiles in pairs (fb 0,1 / 2,3 of a token block): a pair's MFMAs are issued one pair ahead of its 16 VALU instructions
  // (8 multiplies, then 8 FMAs: no dependent back-to-back issue)
  constexpr int GS = 2, NG = 4 * NTB / GS, GPB = 4 / GS, DEPTH = 1;
  v4f_t acc[DEPTH + 1][GS];
  auto mma = [&](int g) {
    const int tb = g / GPB, f0 = (g % GPB) * GS;
#pragma unroll
    for (int k = 0; k < GS; ++k) {
      acc[g % (DEPTH + 1)][k] = v4f_t{0.f, 0.f, 0.f, 0.f};
      if constexpr (C::FS)   // scale operands 0, 0 select the unscaled v_mfma_f32_16x16x128_f8f6f4 (8 % shorter issue)
        acc[g % (DEPTH + 1)][k] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[f0 + k], bf[tb & 1], acc[g % (DEPTH + 1)][k], 3, 3, 0, 0, 0, 0);
      else
        acc[g % (DEPTH + 1)][k] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[f0 + k], bf[tb & 1], acc[g % (DEPTH + 1)][k], 3, 3, 0, 127, 0, 127);
    }
  };
  mma(0);
#ifdef ATOM_TOOLS
  if (tp) { __builtin_amdgcn_sched_barrier(0); tp[0] = (unsigned)__builtin_amdgcn_s_memtime(); }   // first MFMAs issued: fragments arrived
#endif
  float sa = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int tb = g / GPB, f0 = (g % GPB) * GS;
    __builtin_amdgcn_sched_barrier(0);
#ifdef ATOM_TOOLS
    if (tp && g == NG / 2) tp[1] = (unsigned)__builtin_amdgcn_s_memtime();
#endif
    if (g % GPB == 0) {
      sa = (float)sah;
      if (tb + 1 < NTB) {
        bf[(tb + 1) & 1] = frag24(pa + (tb + 1) * 16 * PITCH);
        sah = *reinterpret_cast<const sa_t *>(psa + SA_AT + (tb + 1) * 16 * PITCH);
      }
      // the two waves of a SIMD swap priority mid-step (see compute_int4)
      if constexpr (PH) {
        if (tb == NTB / 2) __builtin_amdgcn_s_barrier();
      }
      if constexpr (PRIO) {
        if (tb == 0) { if (older) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2); }
        if (tb == NTB / 2) { if (older) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
      }
    }
    if (g + DEPTH < NG) mma(g + DEPTH);
    if (g % GPB == GPB - 1) {
#pragma unroll
      for (int i = (tb * C::GLDS + NTB - 1) / NTB; i < ((tb + 1) * C::GLDS + NTB - 1) / NTB; ++i) dma(i);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      float w[4 * GS];                                       // plain rows: micro-tile k = features 16 (f0 + k) + 4 kb + r
#pragma unroll
      for (int k = 0; k < GS; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (C::FS) w[4 * k + r] = sb[f0 + k][r];
          else w[4 * k + r] = (float)reinterpret_cast<const half_t *>(&sb[f0 + k])[r];
        }
      deq_pair<false, false>(acc[g % (DEPTH + 1)][0], acc[g % (DEPTH + 1)][1], sa, w, c[f0][tb], c[f0 + 1][tb]);
    }
  }
}

// the keeper on v_mfma_i32_16x16x64_i8, same tile layout: its two 64-column halves sit in two stage slots and are multiplied in ONE
// step -- two chained MFMAs per micro-tile, one de-quantisation (the contract)
template <class C, bool PH = false>
__device__ __forceinline__ void compute_keeper_x16(const char *slot, const char *slot1, int wm, int wn, int lane, float (&c)[4][C::WM / 16][4]) {
  const int l15 = lane & 15, kb = lane >> 4;
  const int sw = (l15 >> 2) & 3;
  const int ow = (wn * 64 + l15) * 64 + ((kb ^ sw) << 4);                              // + fb*16*64
  constexpr int NTB = C::WM / 16;
  const int oa = (C::BN + wm * C::WM + l15) * 64 + ((kb ^ sw) << 4);                   // + tb*16*64
  const char *psa = slot + C::KP_SA_OFF + (wm * C::WM + l15) * 4;                      // + tb*64
  const char *psb = slot + C::SB_OFF + (wn * 64 + 4 * kb) * 2;
  v4i af[4], af1[4];
  v2u sb[4];
#pragma unroll
  for (int fb = 0; fb < 4; ++fb) {
    af[fb] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot + ow + fb * 1024));
    af1[fb] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot1 + ow + fb * 1024));
    sb[fb] = *reinterpret_cast<const v2u *>(psb + fb * 32);
  }
#pragma unroll
  for (int tb = 0; tb < NTB; ++tb) {
    if constexpr (PH) {
      if (tb == NTB / 2) __builtin_amdgcn_s_barrier();
    }
    const v4i b = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot + oa + tb * 1024));
    const v4i b1 = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot1 + oa + tb * 1024));
    const float sa = (float)*reinterpret_cast<const half_t *>(psa + tb * 64);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
      v4i a = {0, 0, 0, 0};
      a = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[fb], b, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_i32_16x16x64_i8(af1[fb], b1, a, 0, 0, 0);
      const v4f_t f = {(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
      dequant4x(f, sa, sb[fb], c[fb][tb]);
    }
  }
}

// Two K groups of a workgroup (KG = 2): each group finishes half of the token blocks -- it hands the other half of its FP32 sums to
// its partner wave through LDS, adds the partner's (first half of the K steps + second half) and writes its rows as fp16.
template <class C, int KG>
__device__ __forceinline__ void kg_exchange_store(const GemmParams &p, char *lds_all, float (&c)[4][C::WM / 16][4], int kg, int wave, int wm,
                                                  int wn, int lane, int m0, int n0) {
  constexpr int NTB = C::WM / 16, EP_STRIDE = 144;
  const int l15 = lane & 15, kb = lane >> 4;
  const int wave_all = kg * C::NW + wave;
  constexpr int HB = NTB / 2, XF = HB * 16;             // floats per lane handed over
  float *xw = reinterpret_cast<float *>(lds_all + C::NW * KG * (HB * 16 * EP_STRIDE)) + (kg * C::NW + wave) * XF * 64 + lane;
  const float *xr = reinterpret_cast<const float *>(lds_all + C::NW * KG * (HB * 16 * EP_STRIDE)) + ((1 - kg) * C::NW + wave) * XF * 64 + lane;
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (h != kg) {
#pragma unroll
      for (int t = 0; t < HB; ++t)
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
          for (int r = 0; r < 4; ++r) xw[((t * 4 + fb) * 4 + r) * 64] = c[fb][h * HB + t][r];
    }
  __builtin_amdgcn_s_barrier();
  char *ep = lds_all + wave_all * (HB * 16 * EP_STRIDE);
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (h == kg) {
#pragma unroll
      for (int t = 0; t < HB; ++t)
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
          v2u o;
          half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float other = xr[((t * 4 + fb) * 4 + r) * 64];
            ov[r] = f2h(kg == 0 ? c[fb][h * HB + t][r] + other : other + c[fb][h * HB + t][r]);
          }
          *reinterpret_cast<v2u *>(ep + (t * 16 + l15) * EP_STRIDE + (fb * 16 + 4 * kb) * 2) = o;
        }
    }
#pragma unroll
  for (int i = 0; i < HB * 2; ++i) {
    const int rl = i * 8 + (lane >> 3);
    const int ch = lane & 7;
    const v4u v = *reinterpret_cast<const v4u *>(ep + rl * EP_STRIDE + ch * 16);
    const int m = m0 + wm * C::WM + kg * (HB * 16) + rl;
    const int n = n0 + wn * 64 + ch * 8;
    if (m < p.M && n < p.N) *reinterpret_cast<v4u *>(p.D + (int64_t)m * p.N + n) = v;
  }
}

// Four K groups (KG = 4; 64x128 tiles, wave tiles of 4 feature blocks x 2 token blocks = 8 micro-tiles): group g finishes micro-tiles
// 2g and 2g + 1 (token block g / 2, feature blocks 2 (g % 2) + {0, 1}); every wave parks the six it does not own in LDS and the owner
// adds the four partial sums in K order, ((p0 + p1) + p2) + p3.
template <class C>
__device__ __forceinline__ void kg4_exchange_store(const GemmParams &p, char *lds_all, float (&c)[4][C::WM / 16][4], int kg, int wave, int wm,
                                                   int wn, int lane, int m0, int n0) {
  static_assert(C::WM == 32, "kg4: wave tiles of 64 features x 32 tokens");
  constexpr int EP_STRIDE = 80, EP_WAVE = 16 * EP_STRIDE, NWA = C::NW * 4;
  const int l15 = lane & 15, kb = lane >> 4;
  float *xb = reinterpret_cast<float *>(lds_all + NWA * EP_WAVE);          // [group][wave][micro-tile][r][lane]
  auto xat = [&](int g, int mt, int r) { return xb + ((((g * C::NW + wave) * 8 + mt) * 4 + r) << 6) + lane; };
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
    if (mt >> 1 != kg) {
#pragma unroll
      for (int r = 0; r < 4; ++r) *xat(kg, mt, r) = c[mt & 3][mt >> 2][r];
    }
  __builtin_amdgcn_s_barrier();
  char *ep = lds_all + (kg * C::NW + wave) * EP_WAVE;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    if (g == kg) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int mt = 2 * g + f;
        v2u o;
        half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sum = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float part = k == g ? c[mt & 3][mt >> 2][r] : *xat(k, mt, r);
            sum = k == 0 ? part : sum + part;
          }
          ov[r] = f2h(sum);
        }
        *reinterpret_cast<v2u *>(ep + l15 * EP_STRIDE + (f * 16 + 4 * kb) * 2) = o;
      }
    }
  const int row = lane >> 2, ch = lane & 3;
  const v4u v = *reinterpret_cast<const v4u *>(ep + row * EP_STRIDE + ch * 16);
  const int m = m0 + wm * C::WM + (kg >> 1) * 16 + row;
  const int n = n0 + wn * 64 + (kg & 1) * 32 + ch * 8;
  if (m < p.M && n < p.N) *reinterpret_cast<v4u *>(p.D + (int64_t)m * p.N + n) = v;
}

// KG = 2: two groups of C::NW waves share the tile and split its K steps in halves (each group with its own LDS ring); the
// halves are added through the LDS at the end, lower K range first -- the arithmetic of the split-K route with two splits,
// without its FP32 round trip through HBM.  For shapes with at most one 128x128 tile per CU: the second wave per SIMD hides the
// barrier and fragment-load latency of the first.
template <class C, bool SK = false, int KG = 1, bool PH = false, bool TR = false>
__global__ __launch_bounds__(C::NT * KG, C::OCC) void gemm_w4a4_f6x16_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds_all[];
  constexpr int NS = C::NS;
  constexpr int NTB = C::WM / 16;
  static_assert(C::WM % 32 == 0 && NS >= 2, "x16: wave tiles of 64 features x 32, 64 or 128 tokens");
  static_assert(KG == 1 || ((KG == 2 || KG == 4) && !SK && NTB % 2 == 0), "K groups: two or four, without the global split");
  static_assert(!PH || KG == 2, "the half-step phase shift is between two K groups");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave_all >= 0 && wave_all < C::NW * KG);
  const int kg = KG > 1 ? wave_all / C::NW : 0;
  const int wave = KG > 1 ? wave_all % C::NW : wave_all;
  char *lds = lds_all + kg * (NS * C::STAGE_BYTES);
  const int wm = wave / C::WGN, wn = wave % C::WGN;
  const int nbn = (p.N + C::BN - 1) / C::BN, nbm = (p.M + C::BM - 1) / C::BM;
  const int nwg = nbm * nbn;
  int id = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  constexpr int GM = 4;
  const int band = id / (GM * nbn), inband = id % (GM * nbn);
  const int rows_in_band = min(GM, nbm - band * GM);
  const int bm = band * GM + inband % rows_in_band, bn = inband / rows_in_band;
  const int m0 = bm * C::BM, n0 = bn * C::BN;

  float c[4][NTB][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NTB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[a][b][r] = 0.f;

  // K steps: the G int4 groups, then the keeper = G + 1 compute steps; DMA stages: G + 2 (the keeper's two 64-column halves).
  // split-K (SK): blockIdx.y owns the compute steps [s_begin, c_end) and writes FP32 partial sums to p.ws; KG > 1: group kg owns
  // [csteps kg / KG, csteps (kg + 1) / KG)
  const int csteps = p.G + 1;
  const int s_begin = SK ? (int)((int64_t)csteps * blockIdx.y / p.splits) : (KG > 1 ? csteps * kg / KG : 0);
  const int c_end = SK ? (int)((int64_t)csteps * (blockIdx.y + 1) / p.splits) : (KG > 1 ? csteps * (kg + 1) / KG : csteps);
  const bool has_keeper = c_end == csteps;
  const int nsteps = c_end + (has_keeper ? 1 : 0);          // DMA stages [s_begin, nsteps)
  auto issue = [&](int step) {
    char *slot = lds + (step % NS) * C::STAGE_BYTES;
    const int s = min(step, nsteps - 1);
    if (s < p.G) issue_int4<C>(p, s, slot, wave, lane, m0, n0);
    else issue_keeper<C>(p, s - p.G, slot, wave, lane, m0, n0);
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s_begin + s);
  const bool older = KG > 1 ? kg == 0 : wave < C::NW / 2;
  constexpr bool PRIO = KG > 1 || C::NW >= 8;
  // tools build, TR: s_memtime stamps of workgroup 0, 64 dwords per wave in the LDS behind the stages, copied to p.Dsz at the
  // end: [0] entry, [1] first stages issued, [2 + 6 j + k] step j < 10 of the wave's range: k = 0 loop top, 1 stage landed
  // (vmcnt), 2 behind the barrier, 3 first MFMAs issued, 4 half of the tiles done, 5 step done; [62] loop done, [63] end
  unsigned *trl = nullptr;
  if constexpr (TR) {
    if (blockIdx.x == 0) trl = reinterpret_cast<unsigned *>(lds_all + KG * NS * C::STAGE_BYTES) + wave_all * 64;
  }
  auto stamp = [&](int k) {
    if constexpr (TR) {
      if (trl) { __builtin_amdgcn_sched_barrier(0); const unsigned t = (unsigned)__builtin_amdgcn_s_memtime(); if (lane == 0) trl[k] = t; __builtin_amdgcn_sched_barrier(0); }
    }
  };
  stamp(1);
  if constexpr (PH) {
    if (kg == 1) __builtin_amdgcn_s_barrier();             // the second group runs half a step behind
  }
  int step = s_begin;
  for (; step + NS - 1 < min(p.G, nsteps); ++step) {
    const bool tr_on = TR && trl && step - s_begin < 10;
    const int tb_ = 2 + 6 * (step - s_begin);
    if (tr_on) stamp(tb_);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::GLDS * (NS - 2)) : "memory");
    if (tr_on) stamp(tb_ + 1);
    __builtin_amdgcn_s_barrier();
    if (tr_on) stamp(tb_ + 2);
    char *nslot = lds + ((step + NS - 1) % NS) * C::STAGE_BYTES;
    const int g = step + NS - 1;
    auto dma = [&](int i) { issue_int4_piece<C>(p, g, nslot, wave, lane, m0, n0, i); };
    __builtin_amdgcn_sched_barrier(0);
    unsigned tmp2[2] = {0, 0};
    compute_int4_x16<C, decltype(dma), PRIO, PH>(lds + (step % NS) * C::STAGE_BYTES, wm, wn, lane, c, dma, older, tr_on ? tmp2 : nullptr);
    if (tr_on) { if (lane == 0) { trl[tb_ + 3] = tmp2[0]; trl[tb_ + 4] = tmp2[1]; } stamp(tb_ + 5); }
  }
  stamp(62);
  for (; step < min(p.G, nsteps); ++step) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::GLDS * (NS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(step + NS - 1);
    __builtin_amdgcn_sched_barrier(0);
    compute_int4_x16<C, NoDma, PRIO, PH>(lds + (step % NS) * C::STAGE_BYTES, wm, wn, lane, c, NoDma(), older);
  }
  __builtin_amdgcn_s_setprio(0);
  if (has_keeper) {                                        // step == G: both halves (stages G, G + 1) in one compute step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (NS == 2) {                               // two-stage ring: the second half only fits once stage G - 1 is read
      issue_keeper<C>(p, 1, lds + ((p.G + 1) % NS) * C::STAGE_BYTES, wave, lane, m0, n0);   // (one more barrier: the K groups below
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     //  count it in)
      __builtin_amdgcn_s_barrier();
    }
    compute_keeper_x16<C, PH>(lds + (p.G % NS) * C::STAGE_BYTES, lds + ((p.G + 1) % NS) * C::STAGE_BYTES, wm, wn, lane, c);
    ++step;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int l15 = lane & 15, kb = lane >> 4;
  constexpr int EP_STRIDE = 144;
  if constexpr (PH) {
    if (kg == 0) __builtin_amdgcn_s_barrier();
  }
  if constexpr (KG > 1) {
    // one barrier per compute step above (two with the half-step phase shift; one more in the keeper's group of a two-stage ring):
    // the groups with fewer catch up
    for (int i = (c_end - s_begin) * (PH ? 2 : 1) + (NS == 2 && has_keeper ? 1 : 0); i < (csteps + KG - 1) / KG * (PH ? 2 : 1) + (NS == 2 ? 1 : 0); ++i)
      __builtin_amdgcn_s_barrier();
    if constexpr (KG == 4) kg4_exchange_store<C>(p, lds_all, c, kg, wave, wm, wn, lane, m0, n0);
    else kg_exchange_store<C, KG>(p, lds_all, c, kg, wave, wm, wn, lane, m0, n0);
    if constexpr (TR) {
      if (trl) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(63);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        reinterpret_cast<unsigned *>(p.Dsz)[wave_all * 64 + lane] = trl[lane];
      }
    }
    return;
  }
  if constexpr (SK) {                                     // FP32 partial tile: 4 consecutive features per lane and micro-tile
    float *wsp = p.ws + (int64_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
      const int m = m0 + wm * C::WM + tb * 16 + l15;
      if (m >= p.M) continue;
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) {
        const int n = n0 + wn * 64 + fb * 16 + 4 * kb;
        if (n >= p.N) continue;
        *reinterpret_cast<v4f *>(wsp + (int64_t)m * p.N + n) = v4f{c[fb][tb][0], c[fb][tb][1], c[fb][tb][2], c[fb][tb][3]};
      }
    }
    return;
  }
  // epilogue: per wave [HT tokens][64 features] fp16 through LDS (row stride 144 B), HT = 64 (32 for the 32-token wave tile)
  constexpr int HT = C::WM < 64 ? C::WM : 64;
  char *ep = lds + wave * (HT * EP_STRIDE);
#pragma unroll
  for (int half = 0; half < C::WM / HT; ++half) {
#pragma unroll
    for (int t4 = 0; t4 < HT / 16; ++t4) {
      const int tb = half * (HT / 16) + t4;
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) {
        v2u o;
        half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
        for (int k = 0; k < 4; ++k) ov[k] = f2h(c[fb][tb][k]);
        *reinterpret_cast<v2u *>(ep + (t4 * 16 + l15) * EP_STRIDE + (fb * 16 + 4 * kb) * 2) = o;
      }
    }
#pragma unroll
    for (int i = 0; i < HT / 8; ++i) {
      const int rl = i * 8 + (lane >> 3);
      const int ch = lane & 7;
      const v4u v = *reinterpret_cast<const v4u *>(ep + rl * EP_STRIDE + ch * 16);
      const int m = m0 + wm * C::WM + half * HT + rl;
      const int n = n0 + wn * 64 + ch * 8;
      if (m < p.M && n < p.N) *reinterpret_cast<v4u *>(p.D + (int64_t)m * p.N + n) = v;
    }
  }
}

// ================================================================================================================
// The 256x256 product kernel, second generation ("p" = pipelined across K steps).  Same stage layout, DMA, arithmetic
// contract and output as gemm_w4a4_f6x16_kernel; what changed follows from tools/probes/issue_probe (profiles/r02):
//  * MFMA and VALU share a SIMD's issue port: one 16x16x128 BF6 MFMA hides ~2 VALU instructions, every further one costs
//    ~2.4 cycles whatever the number of waves -- the loop is bound by MFMA + VALU issue, so everything that is not one of
//    the 8 de-quantisation instructions per micro-tile had to go: the weight scales are converted to FP32 once per step
//    (plain v_fma_f32: v_fma_mix_f32 costs 1.5x a v_fma_f32), and the MFMA is the UNSCALED v_mfma_f32_16x16x128_f8f6f4
//    (the scaled form with unit scales is 10 % slower: 18.5 vs 16.8 cycles back to back).
//  * One s_barrier per K step, in the MIDDLE of the step: it publishes stage s+1 (landed since the previous step) and
//    releases slot (s+2) % 3 for the next LDS-DMA.  Nothing happens at the step boundary any more -- the first fragments of
//    stage s+1 are read during the tail of step s (the r01 kernel idled ~500 of 4500 cycles per step between its barrier
//    and the first MFMA while all 8 waves hit the LDS at once).
//  * Fragment rows are interleaved: micro-tile row i of fragment f of a 32-row block is data row 2i + (f & 1).  With the
//    104-byte pitch the 32 ds_read_b64 of a lane group then fall into 32 distinct bank pairs (consecutive rows collide
//    once per group: 26*15 + 0 == 26*0 + 6 (mod 64) -- SQ_LDS_BANK_CONFLICT 2.1 M per launch in r01), each fragment is
//    three ds_read_b64 (the compiler's ds_read2_b64 runs at half the LDS rate), and a lane ends up with 8 CONSECUTIVE
//    output features per token: the epilogue is one 16-byte global store per lane and feature-block pair, no LDS pass.
//  * Token fragments are re-loaded as soon as the last MFMA reading the register has issued (3 pair-slots ahead of their
//    use instead of 1); block 0 of the next step has a buffer of its own.
// Wave tile: 64 features (4 blocks fb) x 128 tokens (8 blocks tb); lane l: micro-tile row / column l % 16, k-block l / 16.
template <class C>
struct PRegs {
  v8i af[4];            // feature fragments of the current stage
  v8i bf[3];            // token fragments: [2] block 0, [0] blocks 2,4,6, [1] odd blocks
  float sb[4][4];       // FP32 weight scales [fb][r]
  v4u sbp[2];           // the next stage's weight scales, packed fp16 (8 consecutive features per fb pair)
  float sa[8];          // token scales of the current step (converted on arrival)
  half_t sah[3];        // token scales in flight, by token-fragment buffer
  v4f_t acc[2][2];
  int lo[3];            // lane offset of a fragment's three 8-byte pieces; [1], [2] opaque to the compiler (see frag3)
};

// 24 bytes of a row as three ds_read_b64.  The three addresses come from three registers the compiler cannot relate to each
// other (lo[1], lo[2] went through an empty asm), otherwise it merges two of the loads into a ds_read2_b64, which runs at
// half the LDS rate (MI355X_MICROARCH.md, LDS table).
template <class C>
__device__ __forceinline__ v8i frag3(const PRegs<C> &R, const char *base, int off) {
  const v2u a = *reinterpret_cast<const v2u *>(base + R.lo[0] + off);
  const v2u b = *reinterpret_cast<const v2u *>(base + R.lo[1] + off);
  const v2u c = *reinterpret_cast<const v2u *>(base + R.lo[2] + off);
  asm volatile("" ::: "memory");   // ... and keeps it from pairing this fragment's loads with the next fragment's (same registers)
  return v8i{(int)a.x, (int)a.y, (int)b.x, (int)b.y, (int)c.x, (int)c.y, 0, 0};
}

template <class C>
__device__ __forceinline__ v8i frag3_untracked(const PRegs<C> &R, const char *base, int off) {   // tools only (ABL & 256)
  v2u a, b, c;
  const unsigned a0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(base + R.lo[0] + off);
  const unsigned a1 = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(base + R.lo[1] + off);
  const unsigned a2 = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(base + R.lo[2] + off);
  asm volatile("s_waitcnt lgkmcnt(8)\n\tds_read_b64 %0, %3\n\tds_read_b64 %1, %4\n\tds_read_b64 %2, %5" : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(a0), "v"(a1), "v"(a2));
  return v8i{(int)a.x, (int)a.y, (int)b.x, (int)b.y, (int)c.x, (int)c.y, 0, 0};
}

// pair slot i of a step -> token block, feature-block pair (0 = fb 0,1; 1 = fb 2,3).  Blocks 6 and 7 run A, A, B, B so that
// the registers of fragments fb 0,1 are free two slots before the step ends.
__device__ __forceinline__ constexpr int p_tb(int i) { return i < 12 ? i / 2 : (i == 12 || i == 14 ? 6 : 7); }
__device__ __forceinline__ constexpr int p_h(int i) { return i < 12 ? i % 2 : (i >= 14 ? 1 : 0); }
__device__ __forceinline__ constexpr int p_buf(int tb) { return tb == 0 ? 2 : (tb & 1); }
__device__ __forceinline__ constexpr int p_row(int blk) { return 32 * (blk >> 1) + (blk & 1); }   // + 2 * (l % 16)

template <class C>
__device__ __forceinline__ void p_load_sb(PRegs<C> &R, const char *slot, int wn, int kb) {
  const char *psb = slot + C::SB_OFF + (wn * 64 + 8 * kb) * 2;
  R.sbp[0] = *reinterpret_cast<const v4u *>(psb);
  R.sbp[1] = *reinterpret_cast<const v4u *>(psb + 64);
}
template <class C>
__device__ __forceinline__ void p_cvt_sb(PRegs<C> &R, int h) {   // feature 8kb + 2r + (fb & 1) of pair h -> sb[2h + (fb & 1)][r]
  const half_t *hv = reinterpret_cast<const half_t *>(&R.sbp[h]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    R.sb[2 * h][r] = (float)hv[2 * r];
    R.sb[2 * h + 1][r] = (float)hv[2 * r + 1];
    asm volatile("" : "+v"(R.sb[2 * h][r]), "+v"(R.sb[2 * h + 1][r]));   // opaque: else the conversion is folded back into
  }                                                                        // v_fma_mix_f32 (1.5x the cost of v_fma_f32)
}

// One K step (an int4 group) of the pipelined kernel.  LAST: no next int4 stage to prefetch from.  `mid(i)` is called behind
// the MFMAs of slots 8..15 with i = 0..7 (the caller issues its LDS-DMA there); `sync()` is the mid-step wait + barrier.
// Slot 0 de-quantises the pair carried over from the previous step (all-zero registers in the first step).
// ABL (tools build only): 1 = no LDS-DMA after the prologue, 2 = no de-quantisation, 4 = no MFMA, 8 = no fragment / scale
// re-loads, 16 = s_memtime stamps through `stamp(k)`, 64 = no output stores, 128 = no keeper steps
template <class C, bool LAST, int ABL, class FS, class FD, class FT>
__device__ __forceinline__ void p_step(PRegs<C> &R, const char *slot, const char *nslot, int wm, int wn, int lane,
                                       float (&c)[4][8][4], FS sync, FD mid, FT stamp) {
  const int l15 = lane & 15, kb = lane >> 4;
  const bool older = wm == 0;                               // waves 0-3 (the first wave of each SIMD)
  const char *pa = slot + C::A_OFF + wm * 128 * PITCH;
  const char *npw = nslot + wn * 64 * PITCH, *npa = nslot + C::A_OFF + wm * 128 * PITCH;
  const char *psa = pa + l15 * (2 * PITCH) + 96, *npsa = npa + l15 * (2 * PITCH) + 96;
  // ABL & 256 (tools): the same LDS reads into the same registers, but issued through inline asm the compiler does not track, so
  // nothing ever waits for them (a guard keeps <= 12 in flight): prices the s_waitcnt stalls of the fragment prefetch.  Results
  // are garbage.
#define ATOM_P_LOADF(dst, base, off) { if constexpr ((ABL & 256) != 0) dst = frag3_untracked<C>(R, base, off); else dst = frag3<C>(R, base, off); }
#define ATOM_P_LOADH(dst, ptr) { if constexpr ((ABL & 256) != 0) { unsigned t_; asm volatile("ds_read_u16 %0, %1" : "=v"(t_) : "v"((unsigned)(size_t)(const __attribute__((address_space(3))) char *)(ptr))); dst = __builtin_bit_cast(half_t, (unsigned short)t_); } else dst = *reinterpret_cast<const half_t *>(ptr); }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int tb = p_tb(i), h = p_h(i);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((ABL & 16) != 0) { if (i == 0) stamp(0); if (i == 4) stamp(1); if (i == 12) stamp(4); }
    if (i == 8) {
      if constexpr ((ABL & 16) != 0) stamp(2);
      sync();
      if constexpr ((ABL & 16) != 0) stamp(3);
    }
    if (h == 0) R.sa[tb] = (float)R.sah[p_buf(tb)];         // this block's token scale arrived with its fragment
    if constexpr ((ABL & 2048) != 0) {                      // tools: the two waves of a SIMD swap priority mid-step (as in the x16 kernel)
      if (i == 0) { if (older) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2); }
      if (i == 8) { if (older) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
    }
    if constexpr ((ABL & 4096) != 0) {                      // tools: ... the other way round
      if (i == 0) { if (older) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
      if (i == 8) { if (older) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2); }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if constexpr (!(ABL & 4))
        R.acc[i & 1][k] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(R.af[2 * h + k], R.bf[p_buf(tb)], v4f_t{0.f, 0.f, 0.f, 0.f}, 3, 3, 0, 0, 0, 0);
      else { R.acc[i & 1][k] = v4f_t{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+v"(R.acc[i & 1][k]) : "v"(R.af[2 * h + k]), "v"(R.bf[p_buf(tb)])); }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- loads behind this slot's MFMAs, into registers whose last reader has just issued
    if (!(ABL & 8) && h == 1 && i < 12 && tb + 2 < 8) {      // token block tb + 2 (the buffer of block tb; block 0: of block 6
      ATOM_P_LOADF(R.bf[p_buf(tb + 2)], pa, p_row(tb + 2) * PITCH)   // of the previous step)
      ATOM_P_LOADH(R.sah[p_buf(tb + 2)], psa + p_row(tb + 2) * PITCH)
    }
    if constexpr (!LAST && !(ABL & 8)) {
      if (i == 10) p_load_sb<C>(R, nslot, wn, kb);
      if (i == 12) {                                         // next step's block 0 (buffer 2: free since slot 1)
        ATOM_P_LOADF(R.bf[2], npa, p_row(0) * PITCH)
        ATOM_P_LOADH(R.sah[2], npsa + p_row(0) * PITCH)
      }
      if (i == 13) {                                         // fragments fb 0,1: last read by this slot
        ATOM_P_LOADF(R.af[0], npw, p_row(0) * PITCH)
        ATOM_P_LOADF(R.af[1], npw, p_row(1) * PITCH)
      }
      if (i == 15) {
        ATOM_P_LOADF(R.af[2], npw, p_row(2) * PITCH)
        ATOM_P_LOADF(R.af[3], npw, p_row(3) * PITCH)
        ATOM_P_LOADF(R.bf[1], npa, p_row(1) * PITCH)        // next step's block 1
        ATOM_P_LOADH(R.sah[1], npsa + p_row(1) * PITCH)
      }
    }
    if (i >= 8 && !(ABL & 1)) mid(i - 8);
    __builtin_amdgcn_sched_barrier(0);
    // ---- de-quantisation of the previous slot's pair (slot 15 of the previous step for i == 0)
    if constexpr ((ABL & 2) != 0) {
      const int j = (i + 15) & 15, dtb = p_tb(j), dh = p_h(j);
      c[2 * dh][dtb][0] += R.acc[j & 1][0][0] + R.acc[j & 1][1][3];
      asm volatile("" : "+v"(c[2 * dh][dtb][0]) : "v"(R.acc[j & 1][0]), "v"(R.acc[j & 1][1]));
    } else {
      const int j = (i + 15) & 15, dtb = p_tb(j), dh = p_h(j);
      deq_pair<false, false>(R.acc[j & 1][0], R.acc[j & 1][1], R.sa[dtb], &R.sb[2 * dh][0], c[2 * dh][dtb], c[2 * dh + 1][dtb]);   // sb[2 dh + k][r]
    }
    // the next stage's weight scales replace a pair's FP32 copies once its last de-quantisation of this step is done:
    // pair 0 (slot 13 = block 7) after slot 14's de-quantisation, pair 1 after the carried one in the next step's slot 0
    if constexpr (!LAST) { if (i == 14) p_cvt_sb<C>(R, 0); }
    if (i == 0) p_cvt_sb<C>(R, 1);
  }
  if constexpr ((ABL & 16) != 0) stamp(5);
#undef ATOM_P_LOADF
#undef ATOM_P_LOADH
}

// the carried pair (slot 15) of the last int4 step
template <class C>
__device__ __forceinline__ void p_drain(PRegs<C> &R, float (&c)[4][8][4]) {
  deq_pair<false, false>(R.acc[1][0], R.acc[1][1], R.sa[7], &R.sb[2][0], c[2][7], c[3][7]);
}

// the keeper on v_mfma_i32_16x16x64_i8 with the interleaved row mapping: both 64-column halves (two stage slots) in one step, two
// chained MFMAs per micro-tile, one de-quantisation (the contract)
template <class C>
__device__ __forceinline__ void p_keeper(const char *slot, const char *slot1, int wm, int wn, int lane, float (&c)[4][8][4]) {
  const int l15 = lane & 15, kb = lane >> 4;
  const int sw = (l15 >> 1) & 3;                                       // swizzle key (row >> 2) & 3 of row 2 * l15 + (blk & 1)
  const int ow = (wn * 64 + 2 * l15) * 64 + ((kb ^ sw) << 4);
  const int oa = (C::BN + wm * 128 + 2 * l15) * 64 + ((kb ^ sw) << 4);
  const char *psa = slot + C::KP_SA_OFF + (wm * 128 + 2 * l15) * 4;
  const char *psb = slot + C::SB_OFF + (wn * 64 + 8 * kb) * 2;
  v4i af[4], af1[4];
  v4u sbp[2];
#pragma unroll
  for (int fb = 0; fb < 4; ++fb) {
    af[fb] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot + ow + p_row(fb) * 64));
    af1[fb] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot1 + ow + p_row(fb) * 64));
  }
  sbp[0] = *reinterpret_cast<const v4u *>(psb);
  sbp[1] = *reinterpret_cast<const v4u *>(psb + 64);
#pragma unroll
  for (int tb = 0; tb < 8; ++tb) {
    const v4i b = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot + oa + p_row(tb) * 64));
    const v4i b1 = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot1 + oa + p_row(tb) * 64));
    const float sa = (float)*reinterpret_cast<const half_t *>(psa + p_row(tb) * 4);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
      __builtin_amdgcn_sched_barrier(0);
      v4i a = {0, 0, 0, 0};
      a = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[fb], b, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_i32_16x16x64_i8(af1[fb], b1, a, 0, 0, 0);
      const half_t *hv = reinterpret_cast<const half_t *>(&sbp[fb >> 1]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c[fb][tb][r] = __builtin_fmaf((float)a[r], (float)hv[2 * r + (fb & 1)] * sa, c[fb][tb][r]);
        asm volatile("" : "+v"(c[fb][tb][r]));
      }
    }
  }
}

template <class C, int ABL = 0>
__global__ __launch_bounds__(C::NT, C::OCC) void gemm_w4a4_f6p_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // tools build, ABL & 16: 80 dwords of s_memtime stamps per wave of workgroup 0, kept in the 2,560 bytes of LDS behind the
  // three stages and copied to p.Dsz (u32 [8][80]) at the end.  [0..9] kernel phases, [10 + 6 j + k] stamp k of the j-th traced
  // step, [78], [79] s_memrealtime (100 MHz) at entry and exit
  unsigned *trl = nullptr;
  bool tr_on = false;
  int tr_i = 10;
  auto kstamp = [&](int k) {
    if constexpr ((ABL & 16) != 0) {
      if (blockIdx.x == 0) { const unsigned t = (unsigned)__builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) trl[k] = t; }
    }
  };
  auto stamp = [&](int k) {
    if constexpr ((ABL & 16) != 0) {
      if (tr_on) { const unsigned t = (unsigned)__builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) trl[tr_i + k] = t; }
    }
  };
  if constexpr ((ABL & 16) != 0) {
    trl = reinterpret_cast<unsigned *>(lds + 3 * C::STAGE_BYTES) + (threadIdx.x >> 6) * 80;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) trl[78] = (unsigned)__builtin_amdgcn_s_memrealtime();
    kstamp(0);
  }
  static_assert(C::BM == 256 && C::BN == 256 && C::WM == 128 && C::NS == 3, "p kernel: 256x256, 8 waves of 64 x 128");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < C::NW);
  const int wm = wave / C::WGN, wn = wave % C::WGN;
  const int nbn = (p.N + C::BN - 1) / C::BN, nbm = (p.M + C::BM - 1) / C::BM;
  const int nwg = nbm * nbn;
  int id = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  constexpr int GM = 4;
  const int band = id / (GM * nbn), inband = id % (GM * nbn);
  const int rows_in_band = min(GM, nbm - band * GM);
  const int bm = band * GM + inband % rows_in_band, bn = inband / rows_in_band;
  const int m0 = bm * C::BM, n0 = bn * C::BN;

  float c[4][8][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[a][b][r] = 0.f;

  const int G = p.G;                                        // stages 0..G-1: int4 groups; G, G+1: keeper halves
  auto slot_of = [&](int stage) { return lds + (stage % 3) * C::STAGE_BYTES; };
  auto issue = [&](int stage) {
    if (stage < G) issue_int4<C>(p, stage, slot_of(stage), wave, lane, m0, n0);
    else if (stage < G + 2) issue_keeper<C>(p, stage - G, slot_of(stage), wave, lane, m0, n0);
  };
  issue(0);
  issue(1);
  kstamp(1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::GLDS) : "memory");
  __builtin_amdgcn_s_barrier();
  kstamp(2);
  const int l15 = lane & 15, kb = lane >> 4;
  PRegs<C> R;
  R.lo[0] = l15 * (2 * PITCH) + kb * 24;
  R.lo[1] = R.lo[0] + 8;
  R.lo[2] = R.lo[0] + 16;
  asm volatile("" : "+v"(R.lo[1]));
  asm volatile("" : "+v"(R.lo[2]));
  {
    const char *slot = slot_of(0);
    const char *pw = slot + wn * 64 * PITCH, *pa = slot + C::A_OFF + wm * 128 * PITCH;
    const char *psa = pa + l15 * (2 * PITCH) + 96;
    R.bf[2] = frag3<C>(R, pa, p_row(0) * PITCH);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) R.af[fb] = frag3<C>(R, pw, p_row(fb) * PITCH);
    R.bf[1] = frag3<C>(R, pa, p_row(1) * PITCH);
    R.bf[0] = R.bf[1];
    R.sah[2] = *reinterpret_cast<const half_t *>(psa + p_row(0) * PITCH);
    R.sah[1] = *reinterpret_cast<const half_t *>(psa + p_row(1) * PITCH);
    R.sah[0] = (half_t)0;
    p_load_sb<C>(R, slot, wn, kb);
    p_cvt_sb<C>(R, 0);
    p_cvt_sb<C>(R, 1);
#pragma unroll
    for (int k = 0; k < 2; ++k) R.acc[1][k] = v4f_t{0.f, 0.f, 0.f, 0.f};   // the "carried pair" of the first step: zeros
#pragma unroll
    for (int t = 0; t < 8; ++t) R.sa[t] = 0.f;
  }
  auto sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!(ABL & 512)) __builtin_amdgcn_s_barrier();   // (512, tools: timing without the barrier -- results are garbage)
  };
  // Steps 0 .. G-1.  The LDS-DMA of stage s + 2 goes out behind the mid-step barrier of step s: int4 stages one instruction
  // per pair slot, the keeper halves (the last two stages) in one block.
  {
    int s = 0;
    auto dma4 = [&](int i) { if (i < C::GLDS) issue_int4_piece<C>(p, s + 2, slot_of(s + 2), wave, lane, m0, n0, i); };
    auto dmak0 = [&](int i) { if (i == 0) issue_keeper<C>(p, 0, slot_of(G), wave, lane, m0, n0); };
    auto dmak1 = [&](int i) { if (i == 0) issue_keeper<C>(p, 1, slot_of(G + 1), wave, lane, m0, n0); };
    // (traced steps, tools build: 0, 1, 2, two in the middle, the last three)
    auto tr_sel = [&]() {
      if constexpr ((ABL & 16) != 0) {
        if (tr_on) tr_i += 6;
        tr_on = blockIdx.x == 0 && tr_i + 6 <= 78 && (s < 3 || s == G / 2 || s == G / 2 + 1 || s + 3 >= G);
      }
    };
    kstamp(3);
    if constexpr ((ABL & 1024) != 0) {   // tools: fixed stage slots (what an unrolled-by-3 loop's addressing would cost); results are garbage
      auto dma4f = [&](int i) { if (i < C::GLDS) issue_int4_piece<C>(p, s + 2, lds + 2 * C::STAGE_BYTES, wave, lane, m0, n0, i); };
      for (; s + 2 < G; ++s) p_step<C, false, ABL>(R, lds, lds + C::STAGE_BYTES, wm, wn, lane, c, sync, dma4f, stamp);
    }
    for (; s + 2 < G; ++s) { tr_sel(); p_step<C, false, ABL>(R, slot_of(s), slot_of(s + 1), wm, wn, lane, c, sync, dma4, stamp); }
    if (G >= 2) { tr_sel(); p_step<C, false, ABL>(R, slot_of(s), slot_of(s + 1), wm, wn, lane, c, sync, dmak0, stamp); ++s; }
    tr_sel();
    p_step<C, true, ABL>(R, slot_of(s), slot_of(s + 1), wm, wn, lane, c, sync, dmak1, stamp);
  }
  kstamp(4);
  p_drain<C>(R, c);
  // keeper half 0 was published by the last mid-step barrier; half 1 was issued behind it: both in ONE step once it has landed
  kstamp(5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  kstamp(6);
  if constexpr (!(ABL & 128)) p_keeper<C>(slot_of(G), slot_of(G + 1), wm, wn, lane, c);
  kstamp(7);

  // epilogue: a lane holds 8 consecutive features per token and feature-block pair -> one 16-byte store each
#pragma unroll
  for (int tb = 0; tb < 8; ++tb) {
    const int m = m0 + wm * 128 + p_row(tb) + 2 * l15;
    if (m >= p.M) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = n0 + wn * 64 + 32 * h + 8 * kb;
      if (n >= p.N) continue;
      v4u o;
      half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ov[2 * r] = f2h(c[2 * h][tb][r]);
        ov[2 * r + 1] = f2h(c[2 * h + 1][tb][r]);
      }
      if constexpr ((ABL & 64) != 0) { if (o.x != 0x12345678u) continue; }
      *reinterpret_cast<v4u *>(p.D + (int64_t)m * p.N + n) = o;
    }
  }
  if constexpr ((ABL & 16) != 0) {
    kstamp(8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    kstamp(9);
    if (blockIdx.x == 0) {
      if (lane == 0) trl[79] = (unsigned)__builtin_amdgcn_s_memrealtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (int i = lane; i < 80; i += 64) reinterpret_cast<unsigned *>(p.Dsz)[wave * 80 + i] = trl[i];
    }
  }
}

// ================================================================================================================
// The 256x256 kernel, third generation ("q").  Same tiles, MFMA / de-quantisation stream, arithmetic contract and output as the
// pipelined kernel above; what changed is everything ELSE a K step issues (profiles/r02: on a gfx950 SIMD every instruction's
// pipe time adds up -- MFMA 16 cycles, VALU 2, LDS / DMA / SALU their issue slots -- so the ~190 instructions per wave and step
// that were neither MFMA nor de-quantisation cost ~40 % of the loop):
//  * the K loop is unrolled by 3 = the number of LDS stages, so every stage slot is a compile-time constant: LDS addresses are
//    lane registers set up once + instruction offsets (one register set for slots 0/1, a second one, + 2 stages, for slot 2:
//    ds_read offsets are 16 bits), no `step % 3` arithmetic, no per-load v_add;
//  * token scales are read as the fp32 copy the F6 format carries at byte 100 of a row (no v_cvt), weight scales arrive as
//    fp32 (ATOM_B_F6S: one 1 KiB LDS-DMA piece per stage) and are loaded with ds_read_b128 straight into the registers the
//    de-quantisation FMAs read, at the point where their previous contents die (no packed staging copy, no v_cvt);
//  * LDS-DMA: a wave owns CONSECUTIVE 1 KiB pieces (3 weight + 3 activation, waves 0-4 one more), addressed as SGPR base + one
//    loop-invariant lane offset + instruction offset 0 / 1024 / 2048 -- the offset moves the LDS destination too
//    (tools/probes/dma_offset_probe.cpp) -- so a step computes no DMA address in VALU at all.
template <class C>
struct QC {
  static constexpr int SB_OFF = C::SB_OFF;                  // 256 float32 weight scales (keeper steps: C's fp16 layout)
  static constexpr int STAGE = C::SB_OFF + C::BN * 4;
  static constexpr int LDS_BYTES = 3 * STAGE;
  static_assert(LDS_BYTES <= 160 * 1024 && STAGE + (32 * 3 + 1) * PITCH + 100 < 65536, "q kernel: stage layout / ds offsets");
};

// A token scale in registers.  It arrives as the 8-byte pair at byte 96 of the row's record (fp16 copy, fp32 copy; q_scale below) and
// only the upper dword is used -- but the pair stays ONE live value until its first use: the register allocator otherwise hands the
// dead lower half to the very next temporary, and the write-after-write on a register with a load in flight costs a full
// `s_waitcnt lgkmcnt(0)` behind every scale read (48 of them in the K loop instead of 11; seen in the disassembly, round 6).
typedef float v2f_t __attribute__((ext_vector_type(2)));
#ifdef ATOM_SA_B32
struct QSa {
  float v;
  __device__ __forceinline__ QSa() = default;
  __device__ __forceinline__ QSa(float f) : v(f) {}
  __device__ __forceinline__ operator float() const { return v; }
};
#else
struct QSa {
  v2f_t p;
  __device__ __forceinline__ QSa() = default;
  __device__ __forceinline__ QSa(float f) : p{0.f, f} {}
  __device__ __forceinline__ QSa(v2f_t q) : p(q) {}
  __device__ __forceinline__ operator float() const { asm volatile("" ::"v"(p)); return p.y; }
};
#endif

template <class C, bool PAIR = false>
struct QRegs {
  v8i af[4];            // feature fragments of the current stage
  v8i bf[3];            // token fragments: [2] block 0, [0] blocks 2,4,6, [1] odd blocks
  float sb[2][PAIR ? 4 : 8];   // weight scales of feature-block pair h: [h][2 r + (fb & 1)]; PAIR (channels 2 j, 2 j + 1 share theirs): [h][r]
  QSa sa[4];            // token scales, ring by token block % 4
  v4f_t acc[2][2];
  // lane byte addresses in LDS, [set][piece]: set 0 serves stage slots 0 and 1 (+ instruction offset), set 1 = + 2 stages for
  // slot 2.  All opaque to the compiler: it would otherwise re-derive one from another with a v_add per use, or merge two 8-byte
  // loads of a fragment into a ds_read2_b64 (half the LDS rate).
  int aW[2][3], aA[2][3], aS[2], aB[2];
#ifdef ATOM_F6_PAIR128                       // tools, TIMING ONLY: lane addresses of a pair-interleaved record (see q_frag_pair)
  int aWp[2][3];
#endif
};

// SL = 0..2: compile-time stage slot; SL < 0: the slot's byte offset arrives at run time in `ro` (the two keeper-prefetch steps
// at the end of the K loop, executed once: one code body each instead of one per slot)
template <int SL> __device__ __forceinline__ constexpr int q_set() { return SL == 2 ? 1 : 0; }
template <class C, int SL> __device__ __forceinline__ constexpr int q_imm() { return SL == 1 ? QC<C>::STAGE : 0; }

template <class C, int SL>
__device__ __forceinline__ v8i q_frag(const char *lds, const int (&a)[2][3], int off, int ro) {
  const char *b = lds + (SL < 0 ? ro : q_imm<C, SL>()) + off;
  const v2u x = *reinterpret_cast<const v2u *>(b + a[q_set<SL>()][0]);
  const v2u y = *reinterpret_cast<const v2u *>(b + a[q_set<SL>()][1]);
  const v2u z = *reinterpret_cast<const v2u *>(b + a[q_set<SL>()][2]);
  asm volatile("" ::: "memory");   // keeps the compiler from pairing this fragment's loads with the next fragment's
  return v8i{(int)x.x, (int)x.y, (int)y.x, (int)y.y, (int)z.x, (int)z.y, 0, 0};
}
#ifdef ATOM_F6_PAIR128
// TIMING ONLY (-DATOM_F6_PAIR128, tools/ab_build.sh): the two feature fragments of a row pair as THREE ds_read_b128 -- what a
// pair-interleaved record (row pair i = 208 bytes: per k-block the 16-byte pieces [a0 b0][a1 b1][a2 b2], then the scales) would allow --
// instead of 2 x 3 ds_read_b64.  The bytes in LDS are still the shipped layout, so the RESULTS ARE GARBAGE; the instruction mix, the
// addresses' bank pattern (13 x 16 bytes between lanes: conflict-free) and the waits are those of the real thing.
template <class C, int SL>
__device__ __forceinline__ void q_frag_pair(const char *lds, const int (&a)[2][3], int off, int ro, v8i &f0, v8i &f1) {
  const char *b = lds + (SL < 0 ? ro : q_imm<C, SL>()) + off;
  const v4u x = *reinterpret_cast<const v4u *>(b + a[q_set<SL>()][0]);
  const v4u y = *reinterpret_cast<const v4u *>(b + a[q_set<SL>()][1]);
  const v4u z = *reinterpret_cast<const v4u *>(b + a[q_set<SL>()][2]);
  asm volatile("" ::: "memory");
  f0 = v8i{(int)x.x, (int)x.y, (int)y.x, (int)y.y, (int)z.x, (int)z.y, 0, 0};
  f1 = v8i{(int)x.z, (int)x.w, (int)y.z, (int)y.w, (int)z.z, (int)z.w, 0, 0};
}
#endif
// The token scale of a row: the fp32 copy at byte 100 of its record.  Read as the 8-byte pair at byte 96 (fp16 copy + fp32 copy) and the
// upper dword used (round 6): the 16 rows of a lane group are 208 bytes = 52 dwords apart -- two-way conflicts on the 32 banks a
// ds_read_b32 sees (52 l mod 32 repeats after 8 rows: SQ_LDS_BANK_CONFLICT = 15 % of the LDS cycles of the round-5 kernel), none on the
// 64 a ds_read_b64 sees (52 l mod 64: 16 distinct values) at the same two LDS cycles per instruction.  -DATOM_SA_B32: the round-5 read
// (A/B builds, tools/ab_build.sh).
template <class C, int SL, class RG>
__device__ __forceinline__ QSa q_scale(const char *lds, const RG &R, int off, int ro) {
#ifdef ATOM_SA_B32
  return *reinterpret_cast<const float *>(lds + (SL < 0 ? ro : q_imm<C, SL>()) + off + R.aS[q_set<SL>()] + 4);
#else
  // (a float pair, NOT `bit_cast<float>(v2u.y)`: hipcc of ROCm 7.2 folds the latter to element 0 -- load <2 x float>, extractelement 0 --
  // i.e. the fp16 copy read as a float; found by tests/test_gpu_block.py, kept out by tests/test_abi_cpu.py's disassembly check)
  return QSa(*reinterpret_cast<const v2f_t *>(lds + (SL < 0 ? ro : q_imm<C, SL>()) + off + R.aS[q_set<SL>()]));
#endif
}
template <class C, int SL, class RG>
__device__ __forceinline__ void q_load_sb(const char *lds, RG &R, int h, int ro) {   // 8 consecutive features of pair h
  const char *b = lds + (SL < 0 ? ro : q_imm<C, SL>()) + QC<C>::SB_OFF + 128 * h + R.aB[q_set<SL>()];
  if constexpr (sizeof(R.sb[0]) == 4 * sizeof(float)) {    // PAIR: the even ones (two ds_read2_b32 with dword offsets 0, 2)
    const float *f = reinterpret_cast<const float *>(b);
#pragma unroll
    for (int k = 0; k < 4; ++k) R.sb[h][k] = f[2 * k];
  } else {
    const v4f_t lo = *reinterpret_cast<const v4f_t *>(b), hi = *reinterpret_cast<const v4f_t *>(b + 16);
#pragma unroll
    for (int k = 0; k < 4; ++k) { R.sb[h][k] = lo[k]; R.sb[h][4 + k] = hi[k]; }
  }
}

// LDS-DMA through SGPR base + lane offset + instruction offset (which also moves the LDS destination): three consecutive 1 KiB
// pieces share one M0 and one address; a single piece for the wave's extra one
__device__ __forceinline__ void q_dma16x3(unsigned m0v, unsigned voff, const void *sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:2048" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void q_dma16(unsigned m0v, unsigned voff, const void *sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory");
}
struct QDma {            // per wave, loop invariant
  unsigned voffW, voffA, voffX;   // lane offsets (bytes) from the stage's weight / activation / extra base
  int m0W, m0A, m0X;              // LDS offsets of the wave's first piece inside a stage
  int xkind;                      // extra piece: 1 weight rows, 2 activation rows, 3 weight scales (waves 5-7 repeat the scales piece:
};                                //              same bytes to the same place, and no wave-dependent branch in the K loop)
// DMA group i of this wave (0: its three weight pieces, 3: its three activation pieces, 6: the extra piece; other i: nothing) for
// the stage whose global bases are wsrc / asrc / sbsrc, into stage slot DS
template <class C, int DS>
__device__ __forceinline__ void q_piece(const QDma &d, const uint8_t *wsrc, const uint8_t *asrc, const float *sbsrc, int i) {
  constexpr int base = DS * QC<C>::STAGE;
  if (i == 0) q_dma16x3(base + d.m0W, d.voffW, wsrc);
  else if (i == 3) q_dma16x3(base + d.m0A, d.voffA, asrc);
  else if (i == 6) {
    const void *xb = d.xkind == 1 ? (const void *)wsrc : (d.xkind == 2 ? (const void *)asrc : (const void *)sbsrc);
    q_dma16(base + d.m0X, d.voffX, xb);
  }
}

// One K step on stage slot SL (next stage in slot (SL + 1) % 3; SL < 0: slots at byte offsets ro / rn).  `mid(i)`, i = 0..7:
// behind the MFMAs of slots 8..15.
struct QNoEarly { __device__ __forceinline__ void operator()(int) const {} };
// `early(i)`, i = 0..7: behind the MFMAs of slots 0..7 (step 0 issues the LDS-DMA of stage 1 there; published by the same mid-step barrier)
template <class C, int SL, bool PAIR, class FS, class FD, class FE = QNoEarly>
__device__ __forceinline__ void q_step(QRegs<C, PAIR> &R, const char *lds, float (&c)[4][8][4], FS sync, FD mid, int ro = 0, int rn = 0,
                                       bool LAST = false, FE early = FE(), unsigned *tp = nullptr) {   // LAST (only with SL < 0): no next int4 stage to prefetch from
  // tp (traced tools build, ONE step of two workgroups): s_memtime at the head of every pair slot [i], in front of / behind the mid-step
  // wait + barrier [16], [17], behind the step [18], and around every slot's loads-and-DMA section [32 + 2 i], [33 + 2 i] -- with an
  // s_waitcnt lgkmcnt(0) in front of the second one, so that [33 + 2 i] - [32 + 2 i] is the slot's fragment-read issue + wait + DMA issue
  constexpr int NX = SL < 0 ? -1 : (SL + 1) % 3;
  auto stamp = [&](int k) {
    if (tp) { __builtin_amdgcn_sched_barrier(0); const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) tp[k] = t_; __builtin_amdgcn_sched_barrier(0); }
  };
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int tb = p_tb(i), h = p_h(i);
    __builtin_amdgcn_sched_barrier(0);
    stamp(i);
    if (i == 8) { stamp(16); sync(); stamp(17); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 2; ++k)
      R.acc[i & 1][k] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(R.af[2 * h + k], R.bf[p_buf(tb)], v4f_t{0.f, 0.f, 0.f, 0.f}, 3, 3, 0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    stamp(32 + 2 * i);
    // ---- loads behind this slot's MFMAs, into registers whose last reader has just issued
    if (h == 1 && i < 12 && tb + 2 < 8) {
      R.bf[p_buf(tb + 2)] = q_frag<C, SL>(lds, R.aA, p_row(tb + 2) * PITCH, ro);
      R.sa[(tb + 2) & 3] = q_scale<C, SL>(lds, R, p_row(tb + 2) * PITCH, ro);
    }
    if (!LAST) {
      if (i == 12) {                                         // next step's block 0 (buffer 2: free since slot 1)
        R.bf[2] = q_frag<C, NX>(lds, R.aA, p_row(0) * PITCH, rn);
        R.sa[0] = q_scale<C, NX>(lds, R, p_row(0) * PITCH, rn);
      }
      if (i == 13) {                                         // fragments fb 0,1: last read by this slot
#ifdef ATOM_F6_PAIR128
        q_frag_pair<C, NX>(lds, R.aWp, p_row(0) * PITCH, rn, R.af[0], R.af[1]);
#else
        R.af[0] = q_frag<C, NX>(lds, R.aW, p_row(0) * PITCH, rn);
        R.af[1] = q_frag<C, NX>(lds, R.aW, p_row(1) * PITCH, rn);
#endif
      }
      if (i == 15) {
#ifdef ATOM_F6_PAIR128
        q_frag_pair<C, NX>(lds, R.aWp, p_row(2) * PITCH, rn, R.af[2], R.af[3]);
#else
        R.af[2] = q_frag<C, NX>(lds, R.aW, p_row(2) * PITCH, rn);
        R.af[3] = q_frag<C, NX>(lds, R.aW, p_row(3) * PITCH, rn);
#endif
        R.bf[1] = q_frag<C, NX>(lds, R.aA, p_row(1) * PITCH, rn);        // next step's block 1
        R.sa[1] = q_scale<C, NX>(lds, R, p_row(1) * PITCH, rn);
      }
    }
    if (i >= 8) mid(i - 8);
    else early(i);
    if (tp) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stamp(33 + 2 * i);
    __builtin_amdgcn_sched_barrier(0);
    // ---- de-quantisation of the previous slot's pair (slot 15 of the previous step for i == 0)
    {
      const int j = (i + 15) & 15, dtb = p_tb(j), dh = p_h(j);
      deq_pair<PAIR>(R.acc[j & 1][0], R.acc[j & 1][1], R.sa[dtb & 3], R.sb[dh], c[2 * dh][dtb], c[2 * dh + 1][dtb]);
    }
    // weight scales: pair 0's registers die with slot 13's de-quantisation (done in slot 14) and take the next stage's values;
    // pair 1's die with the carried pair (slot 15, done in the next step's slot 0) and take the current stage's
    __builtin_amdgcn_sched_barrier(0);
    if (!LAST && i == 14) q_load_sb<C, NX>(lds, R, 0, rn);
    if (i == 0) q_load_sb<C, SL>(lds, R, 1, ro);
  }
  stamp(18);
}

// The keeper step of the q kernel (round 3): both 64-column halves (two stage slots) in ONE step, pipelined like q_step.  Round 2 ran
// two half-steps, each MFMA -> convert -> multiply -> FMA per micro-tile with the MFMA's result latency exposed 32 times and a
// barrier between the halves: 4.2 us of the 4096^3 launch for 1/32 of its work, and two de-quantisations where the reference kernel
// does one (Dense_layer_gemm_i4_o16.cuh:640-691).  Now:
//  * pair slots as in q_step -- the four INT8 MFMAs of slot i (two micro-tiles x two chained halves) are issued before the
//    de-quantisation of slot i - 1;
//  * the accumulator starts at the bit pattern of 1.5 * 2^23 (the MFMA's C operand), so that the register read as a float is
//    12582912 + idot exactly: one v_sub per element instead of the quarter-rate v_cvt_f32_i32 (rounds 3-4 folded it into
//    t = fma(acc, sA8, -12582912 * sA8); the round-5 contract multiplies the SCALES first, see deq_pair);
//  * the weight scales are converted to FP32 once, the token fragments of block tb + 2 are requested behind the last MFMA that reads
//    block tb's;
//  * STORE (plain fp16 output): a token block's 64 x 16 outputs are final once its pair h = 1 is de-quantised;
//    they are converted and stored (two 16-byte stores per lane) behind the MFMAs of the following slot, so that the kernel's
//    store tail -- 2-4 us with all 256 workgroups storing 32 MiB at once after the loop -- shrinks to the last block's.
// Same arithmetic as p_keeper: bit-identical results.  Same-box A/B at 4096^3: two pipelined half-steps 56.6 us, this 55.6.
template <class C, bool STORE, int NTB>
__device__ __forceinline__ void q_keeper(const GemmParams &p, const char *slot, const char *slot1, int wm, int wn, int lane, float (&c)[4][NTB][4],
                                         int m0, int n0) {
  constexpr bool MERGED = true;
  const int l15 = lane & 15, kb = lane >> 4;
  const int sw = (l15 >> 1) & 3;                                       // swizzle key (row >> 2) & 3 of row 2 * l15 + (blk & 1)
  const char *pw = slot + (wn * 64 + 2 * l15) * 64 + ((kb ^ sw) << 4);
  const char *pa = slot + (C::BN + wm * (16 * NTB) + 2 * l15) * 64 + ((kb ^ sw) << 4);
  const char *psa = slot + C::KP_SA_OFF + (wm * (16 * NTB) + 2 * l15) * 4;
  const char *psb = slot + C::SB_OFF + (wn * 64 + 8 * kb) * 2;
  v4i af[4], bf[2];
  v4i af1[4], bf1[2];                                      // the second half's fragments (slot1)
  const long d1 = slot1 - slot;
  float sb[2][8], sa[2];
  v4i acc[2][2];
  const v4i magic = {kMagicBits, kMagicBits, kMagicBits, kMagicBits};
#pragma unroll
  for (int fb = 0; fb < 4; ++fb) af[fb] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pw + p_row(fb) * 64));
  bf[0] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pa + p_row(0) * 64));
  bf[1] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pa + p_row(1) * 64));
  if constexpr (MERGED) {
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) af1[fb] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pw + d1 + p_row(fb) * 64));
    bf1[0] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pa + d1 + p_row(0) * 64));
    bf1[1] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pa + d1 + p_row(1) * 64));
  }
  half_t sah[2];
  sah[0] = *reinterpret_cast<const half_t *>(psa + p_row(0) * 4);
  sah[1] = *reinterpret_cast<const half_t *>(psa + p_row(1) * 4);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const v4u w = *reinterpret_cast<const v4u *>(psb + 64 * h);
    const half_t *hv = reinterpret_cast<const half_t *>(&w);
#pragma unroll
    for (int j = 0; j < 8; ++j) { sb[h][j] = (float)hv[j]; asm volatile("" : "+v"(sb[h][j])); }   // opaque: no v_fma_mix re-folding
  }
  auto store_block = [&](int tb) {                                     // fp16 outputs of token block tb: 8 consecutive features per pair
    const int m = m0 + wm * (16 * NTB) + p_row(tb) + 2 * l15;
    if (m >= p.M) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = n0 + wn * 64 + 32 * h + 8 * kb;
      if (n >= p.N) continue;
      v4u o;
      half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ov[2 * r] = f2h(c[2 * h][tb][r]);
        ov[2 * r + 1] = f2h(c[2 * h + 1][tb][r]);
      }
      *reinterpret_cast<v4u *>(p.D + (int64_t)m * p.N + n) = o;
    }
  };
  auto dequant = [&](int j) {                                          // pair slot j: token block j / 2, feature-block pair j % 2
    const int tb = j >> 1, h = j & 1;
    v4f_t f[2];                                                        // the register read as a float is 12582912 + idot: idot exactly
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) f[k][r] = __int_as_float(acc[j & 1][k][r]) - kMagic;
    deq_pair<false>(f[0], f[1], sa[tb & 1], sb[h], c[2 * h][tb], c[2 * h + 1][tb]);   // (the keeper's scales are per output channel, not per pair: model/qLinearLayer.py:59)
  };
#pragma unroll
  for (int i = 0; i < 2 * NTB; ++i) {
    const int tb = i >> 1, h = i & 1;
    __builtin_amdgcn_sched_barrier(0);
    if (h == 0) sa[tb & 1] = (float)sah[tb & 1];                       // this block's token scale arrived with its fragment
#pragma unroll
    for (int k = 0; k < 2; ++k) acc[i & 1][k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[2 * h + k], bf[tb & 1], magic, 0, 0, 0);
    if constexpr (MERGED) {
#pragma unroll
      for (int k = 0; k < 2; ++k) acc[i & 1][k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af1[2 * h + k], bf1[tb & 1], acc[i & 1][k], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (h == 1 && tb + 2 < NTB) {                                        // the buffer of block tb is free: block tb + 2
      bf[tb & 1] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pa + p_row(tb + 2) * 64));
      if constexpr (MERGED) bf1[tb & 1] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pa + d1 + p_row(tb + 2) * 64));
      sah[tb & 1] = *reinterpret_cast<const half_t *>(psa + p_row(tb + 2) * 4);
    }
    if constexpr (STORE) { if (h == 1 && tb >= 1) store_block(tb - 1); }   // final since slot i - 1's de-quantisation
    __builtin_amdgcn_sched_barrier(0);
    if (i > 0) dequant(i - 1);
  }
  __builtin_amdgcn_sched_barrier(0);
  dequant(2 * NTB - 1);
  if constexpr (STORE) store_block(NTB - 1);
}

// GU != 0: the weight rows are gate_proj / up_proj interleaved per wave (32 gate features, then the same 32 up features), so a
// lane ends the K loop with gate and up of the SAME (token, feature) pairs, and the epilogue is the reference's next two ops --
// act_fn(gate) * up and the per-token group quantiser (model/qLlamaLayer.py:345-351; punica/models/llama.py:85-87 ->
// Activate.cuh:67-180) -- writing the F6 activation operand of down_proj directly (GU = 1: simulated-path arithmetic, 2: the CUDA
// kernels').  Bit-identical to fp16 GEMMs + atom_silu_mul_quant_f16; saves writing and re-reading 2 x M x N_inter fp16.
// TR (tools build only, tools/trace_f6q.cpp): s_memtime stamps of workgroups 0 and gridDim.x - 1 into p.Dsz as u32 [2][8 waves][64]:
// [0..15] kernel phases, [16 + s] start of K step s, [62], [63] s_memrealtime (100 MHz) at entry and exit, [64 ..] the in-step stamps of K step 9 (q_step's tp);
// 128 dwords per wave
// PAIR: the caller asserts that output channels 2 j, 2 j + 1 share their weight scales (ATOM_B_SCALE_PAIRS: weight_channel_group = 2,
// the only form the reference kernel accepts): 6 instead of 8 de-quantisation VALU per MFMA (deq_pair), half the scale registers.
template <class C, int GU = 0, bool TR = false, bool PAIR = false>
__global__ __launch_bounds__(C::NT, C::OCC) void gemm_w4a4_f6q_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  static_assert(C::BM == 256 && C::BN == 256 && C::WM == 128 && C::NS == 3 && C::NW == 8, "q kernel: 256x256, 8 waves of 64 x 128");
  using Q = QC<C>;
  unsigned *trb = nullptr;
  auto kstamp = [&](int k) {
    if constexpr (TR) {
      if (trb) { const unsigned t = (unsigned)__builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) trb[k] = t; }
    }
  };
  if constexpr (TR) {
    if (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) {
      trb = reinterpret_cast<unsigned *>(p.Dsz) + ((blockIdx.x ? 8 : 0) + (threadIdx.x >> 6)) * 128;
      if ((threadIdx.x & 63) == 0) trb[62] = (unsigned)__builtin_amdgcn_s_memrealtime();
    }
    kstamp(0);
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < C::NW);
  const int wm = wave / C::WGN, wn = wave % C::WGN;
  const int nbn = (p.N + C::BN - 1) / C::BN, nbm = (p.M + C::BM - 1) / C::BM;
  const int nwg = nbm * nbn;
  int id = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  constexpr int GM = 4;
  const int band = id / (GM * nbn), inband = id % (GM * nbn);
  const int rows_in_band = min(GM, nbm - band * GM);
  const int bm = band * GM + inband % rows_in_band, bn = inband / rows_in_band;
  const int m0 = bm * C::BM, n0 = bn * C::BN;

  float c[4][8][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[a][b][r] = 0.f;

  const int G = p.G;                                        // stages 0..G-1: int4 groups; G, G+1: keeper halves
  // ---- LDS-DMA: consecutive pieces per wave.  Weight rows: pieces 3w..3w+2 (+ 24, 25 on waves 0, 1), activation rows the same
  // (+ 24, 25 on waves 2, 3), the stage's 256 float32 weight scales on wave 4
  QDma d;
  d.voffW = d.voffA = (unsigned)(wave * 3072 + lane * 16);
  d.m0W = wave * 3072;
  d.m0A = C::A_OFF + wave * 3072;
  d.xkind = wave < 2 ? 1 : (wave < 4 ? 2 : 3);
  d.voffX = (unsigned)((wave < 4 ? (24 + (wave & 1)) * 1024 : 0) + lane * 16);
  d.m0X = wave < 2 ? (24 + wave) * 1024 : (wave < 4 ? C::A_OFF + (24 + (wave & 1)) * 1024 : Q::SB_OFF);
  const int64_t wstep = p.f6_rows_b * PITCH, astep = p.f6_rows_a * PITCH;
  const uint8_t *wsrc0 = p.B4 + (int64_t)n0 * PITCH, *asrc0 = p.A4 + (int64_t)m0 * PITCH;
  const float *sbsrc0 = p.sB32 + n0;
  auto slot_of = [&](int stage) { return lds + (stage % 3) * Q::STAGE; };
  auto issue_all = [&](auto ds, int g) {                    // every piece of int4 stage g into slot decltype(ds)::value
    constexpr int DS = decltype(ds)::value;
#pragma unroll
    for (int i = 0; i < 7; ++i) q_piece<C, DS>(d, wsrc0 + g * wstep, asrc0 + g * astep, sbsrc0 + (int64_t)g * p.f6_rows_b, i);
  };
  // Prologue: stage 0 only where the K loop has a regular first step (G >= 6: s = 0 runs on a compile-time-slot body); stage 1's
  // LDS-DMA then goes out behind the MFMAs of that step's first half and is published by its mid-step barrier, where its first
  // reader sits.  (The trace put two stages of DMA issue + landing at 4,500 cycles, the younger wave of each SIMD issuing behind the
  // older one.)
  const bool early1 = G >= 6;
  issue_all(std::integral_constant<int, 0>(), 0);
  if (!early1) {
    if (G >= 2) issue_all(std::integral_constant<int, 1>(), 1);
    else issue_keeper<C>(p, 0, slot_of(1), wave, lane, m0, n0);
  }
  kstamp(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  kstamp(2);

  const int l15 = lane & 15, kb = lane >> 4;
  QRegs<C, PAIR> R;
  {
    const int lw = wn * 64 * PITCH + l15 * (2 * PITCH) + kb * 24;
    const int la = C::A_OFF + wm * 128 * PITCH + l15 * (2 * PITCH);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        R.aW[st][k] = lw + 8 * k + st * 2 * Q::STAGE;
        R.aA[st][k] = la + kb * 24 + 8 * k + st * 2 * Q::STAGE;
        asm volatile("" : "+v"(R.aW[st][k]), "+v"(R.aA[st][k]));
#ifdef ATOM_F6_PAIR128
        R.aWp[st][k] = ((wn * 64 * PITCH + l15 * (2 * PITCH) + kb * 48 + 16 * k + st * 2 * Q::STAGE) & ~15);
        asm volatile("" : "+v"(R.aWp[st][k]));
#endif
      }
      R.aS[st] = la + 96 + st * 2 * Q::STAGE;
      R.aB[st] = (wn * 64 + 8 * kb) * 4 + st * 2 * Q::STAGE;
      asm volatile("" : "+v"(R.aS[st]), "+v"(R.aB[st]));
    }
    R.bf[2] = q_frag<C, 0>(lds, R.aA, p_row(0) * PITCH, 0);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) R.af[fb] = q_frag<C, 0>(lds, R.aW, p_row(fb) * PITCH, 0);
    R.bf[1] = q_frag<C, 0>(lds, R.aA, p_row(1) * PITCH, 0);
    R.bf[0] = R.bf[1];
    R.sa[0] = q_scale<C, 0>(lds, R, p_row(0) * PITCH, 0);
    R.sa[1] = q_scale<C, 0>(lds, R, p_row(1) * PITCH, 0);
    R.sa[2] = 0.f;
    R.sa[3] = 0.f;                                          // the "carried pair" of the first step: zeros x 0
    q_load_sb<C, 0>(lds, R, 0, 0);
    q_load_sb<C, 0>(lds, R, 1, 0);
#pragma unroll
    for (int k = 0; k < 2; ++k) R.acc[1][k] = v4f_t{0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (TR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); kstamp(3); }   // the first fragments have arrived
  auto sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // step s on slot SL: regular (the LDS-DMA of int4 stage s + 2 behind the mid-step barrier), K0 / K1 (the keeper halves instead)
  auto reg = [&](auto sl, int s) {
    constexpr int SL = decltype(sl)::value;
    if constexpr (TR) { if (s < 44) kstamp(16 + s); }
    const int g = s + 2;
    const uint8_t *wsrc = wsrc0 + g * wstep, *asrc = asrc0 + g * astep;
    const float *sbsrc = sbsrc0 + (int64_t)g * p.f6_rows_b;
    unsigned *tp = nullptr;
    if constexpr (TR) { if (trb && s == 9) tp = trb + 64; }   // the in-step stamps of K step 9 (a slot-0 body in the middle of the loop)
    q_step<C, SL, PAIR>(R, lds, c, sync, [&](int i) { q_piece<C, (SL + 2) % 3>(d, wsrc, asrc, sbsrc, i); }, 0, 0, false, QNoEarly(), tp);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  int s = 0;
  if (early1) {                                             // step 0: + the LDS-DMA of stage 1 in its first half
    if constexpr (TR) kstamp(16);
    const uint8_t *wsrc = wsrc0 + 2 * wstep, *asrc = asrc0 + 2 * astep;
    const float *sbsrc = sbsrc0 + (int64_t)2 * p.f6_rows_b;
    q_step<C, 0, PAIR>(R, lds, c, sync, [&](int i) { q_piece<C, 2>(d, wsrc, asrc, sbsrc, i); }, 0, 0, false,
                 [&](int i) { q_piece<C, 1>(d, wsrc0 + wstep, asrc0 + astep, sbsrc0 + p.f6_rows_b, i); });
    reg(S1(), 1);
    reg(S2(), 2);
    s = 3;
  }
  for (; s + 4 < G; s += 3) { reg(S0(), s); reg(S1(), s + 1); reg(S2(), s + 2); }
  // s is a multiple of 3 here and 2..4 steps remain: the ones that still fetch an int4 stage (0..2 of them) are regular steps in
  // slots 0 and 1 -- compile-time bodies again (round 5: ~50 address instructions fewer per wave and step than the generic body) --,
  // the last two, whose LDS-DMA is a keeper half, run on ONE generic code body with run-time stage slots (the second one has no next
  // int4 stage to prefetch from).  (Round 3 tried compile-time slots for those two as well -- a wave-uniform branch per DMA group:
  // ~750 cycles fewer per tail step in the s_memtime trace, no change in wall time at 4096^3 and -0.8 % at 8192^3 under the power
  // cap; profiles/r03_gemm_experiments.txt.)
  if (s + 2 < G) { reg(S0(), s); ++s; }
  if (s + 2 < G) { reg(S1(), s); ++s; }
  for (; s < G; ++s) {
    if constexpr (TR) { if (s < 44) kstamp(16 + s); }
    const int g = s + 2;                                    // the stage this step's LDS-DMA fetches: int4 g, or keeper half g - G
    const int dsl = g % 3;
    const uint8_t *wsrc = wsrc0 + g * wstep, *asrc = asrc0 + g * astep;
    const float *sbsrc = sbsrc0 + (int64_t)g * p.f6_rows_b;
    q_step<C, -1, PAIR>(R, lds, c, sync,
                  [&](int i) {
                    if (g < G) {
                      if (dsl == 0) q_piece<C, 0>(d, wsrc, asrc, sbsrc, i);
                      else if (dsl == 1) q_piece<C, 1>(d, wsrc, asrc, sbsrc, i);
                      else q_piece<C, 2>(d, wsrc, asrc, sbsrc, i);
                    } else if (i == 0) issue_keeper<C>(p, g - G, slot_of(g), wave, lane, m0, n0);
                  },
                  (s % 3) * Q::STAGE, ((s + 1) % 3) * Q::STAGE, s + 1 == G);
  }
  deq_pair<PAIR>(R.acc[1][0], R.acc[1][1], R.sa[3], R.sb[1], c[2][7], c[3][7]);   // the carried pair (slot 15) of the last int4 step
  // keeper half 0 was published by the last mid-step barrier; half 1 was issued behind it: ONE step over both once it has landed
  kstamp(4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  kstamp(5); kstamp(6);
  q_keeper<C, GU == 0>(p, slot_of(G), slot_of(G + 1), wm, wn, lane, c, m0, n0);   // GU == 0: the fp16 output leaves from inside this step
  if constexpr (TR) {
    kstamp(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    kstamp(8);
    if (trb && (threadIdx.x & 63) == 0) trb[63] = (unsigned)__builtin_amdgcn_s_memrealtime();
  }
  if constexpr (GU == 0) return;

  if constexpr (GU != 0) {
    constexpr bool SIM = GU == 1;
    // c[0..1] = gate, c[2..3] = up of features 8 kb + 2 r + (fb & 1) of this wave's 32 (of the tile's 128 = ONE quantisation group)
    float *am = reinterpret_cast<float *>(lds + ((G + 2) % 3) * Q::STAGE);   // [8 waves][128 tokens]: a slot nobody reads any more
#pragma unroll
    for (int tb = 0; tb < 8; ++tb) {
      float mx = 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float h = silu_mul<SIM>((float)f2h(c[k][tb][r]), (float)f2h(c[2 + k][tb][r]));   // on the fp16 GEMM outputs
          c[k][tb][r] = h;
          mx = fmaxf(mx, fabsf(h));
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (kb == 0) am[wave * 128 + p_row(tb) + 2 * l15] = mx;
    }
    __syncthreads();
    const bool keeper_blk = bn == nbn - 1;                  // the last 128 features: INT8 keeper columns
    const GateUpOut &o = p.gu;
    const int N_inter = p.N >> 1;
#pragma unroll
    for (int tb = 0; tb < 8; ++tb) {
      const int t = p_row(tb) + 2 * l15;
      const float *ap = am + wm * 4 * 128 + t;
      const float amax = fmaxf(fmaxf(ap[0], ap[128]), fmaxf(ap[256], ap[384]));
      const GroupScale gs = group_scale<SIM>(amax, keeper_blk, o.clip);
      float tr[8];                                          // feature 8 kb + j: j = 2 r + k
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) tr[2 * r + k] = group_code<SIM>(c[k][tb][r], gs);
      const int64_t m = m0 + wm * 128 + t;
      if (m >= p.M) continue;
      const half_t sh = f2h(gs.s_store);
      if (keeper_blk) {
        const float lo = __builtin_fmaf(tr[1], 256.f, tr[0] + 32896.f), hi = __builtin_fmaf(tr[3], 256.f, tr[2] + 32896.f);
        const float lo2 = __builtin_fmaf(tr[5], 256.f, tr[4] + 32896.f), hi2 = __builtin_fmaf(tr[7], 256.f, tr[6] + 32896.f);
        const v2u w = v2u{((unsigned)lo | ((unsigned)hi << 16)) ^ 0x80808080u, ((unsigned)lo2 | ((unsigned)hi2 << 16)) ^ 0x80808080u};
        *reinterpret_cast<v2u *>(o.o8 + m * kKeeper + wn * 32 + kb * 8) = w;
      } else {
        typedef float v16f __attribute__((ext_vector_type(16)));
        typedef unsigned v6u __attribute__((ext_vector_type(6)));
        v16f ea, eb;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          ea[i] = i < 4 ? tr[2 * i] : 0.f;
          eb[i] = i < 4 ? tr[2 * i + 1] : 0.f;
        }
        const v6u f = cvt_2xpk16_bf6(ea, eb);   // fields 0..7 = my 8 codes: 48 bits
        uint8_t *dst = o.o6 + ((int64_t)bn * o.o6_rows + m) * PITCH;
        uint8_t *d6 = dst + wn * 24 + kb * 6;               // 2-byte aligned: 4 + 2 or 2 + 4 byte stores
        if ((kb & 1) == 0) {
          *reinterpret_cast<unsigned *>(d6) = f[0];
          *reinterpret_cast<unsigned short *>(d6 + 4) = (unsigned short)f[1];
        } else {
          *reinterpret_cast<unsigned short *>(d6) = (unsigned short)f[0];
          *reinterpret_cast<unsigned *>(d6 + 2) = (f[0] >> 16) | (f[1] << 16);
        }
        if (wn == 0 && kb == 0)                             // the next GEMM reads the token scale from the row: fp16 + fp32
          *reinterpret_cast<v2u *>(dst + 96) = v2u{(unsigned)__builtin_bit_cast(unsigned short, sh), __builtin_bit_cast(unsigned, (float)sh)};
      }
      if (wn == 0 && kb == 0) {
        half_t *sd = keeper_blk ? o.s8 : (o.s4 + (int64_t)bn * o.ld4);
        if (o.ref_layout) {
          const int base = ref_scale_index((int)m);
#pragma unroll
          for (int k = 0; k < 4; ++k) sd[base + 2 * k] = sh;
        } else {
          sd[m] = sh;
        }
      }
      if (o.xq) {                                           // code * scale, exact in FP32, one rounding (see quant_kernels.hip)
        v4u q;
        half_t *qv = reinterpret_cast<half_t *>(&q);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = (half_t)__builtin_fmaf(tr[j], gs.s_dq, 0.0f);
        *reinterpret_cast<v4u *>(o.xq + m * N_inter + bn * 128 + wn * 32 + kb * 8) = q;
      }
    }
    return;
  }
}

template <class C, int GU = 0, bool TR = false, bool PAIR = false>
static int launch_q(const GemmParams &p, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  if (ensure_max_lds(reinterpret_cast<const void *>(&gemm_w4a4_f6q_kernel<C, GU, TR, PAIR>), QC<C>::LDS_BYTES, attr_done) != ATOM_OK) return ATOM_ERR_LAUNCH;
  const int nbm = (p.M + C::BM - 1) / C::BM, nbn = (p.N + C::BN - 1) / C::BN;
  hipLaunchKernelGGL((gemm_w4a4_f6q_kernel<C, GU, TR, PAIR>), dim3((unsigned)(nbm * nbn)), dim3(C::NT), QC<C>::LDS_BYTES, s, p);
  return check_launch();
}

// ================================================================================================================
// The K-group kernel, second generation ("qk"): a 128x128 tile shared by TWO groups of 4 waves (wave tiles 64 features x 64 tokens),
// each group running the q kernel's K step -- compile-time stage slots, fp32 scales from LDS straight into the FMA operands, MFMA of
// pair slot i issued ahead of the de-quantisation of slot i - 1, address-free LDS-DMA -- on its own 3-stage ring over its own range
// of the G + 1 compute steps: group k owns [(G + 1) k / 2, (G + 1) (k + 1) / 2), so the result is the x16 K-group kernel's (and
// atom_gemm_w4a4_f6_order's nsplit = 2) bit for bit.  For shapes with at most one 128x128 tile per CU (512..1024 tokens at N = 4096):
// the x16 body spent 2,950 cycles per wave and K step on 16 MFMAs.
//  * stage = 13 + 13 KiB of rows + 128 float32 weight scales (Cfg<128,128,2,3,3,1>'s layout: the keeper stages and its issue_keeper
//    are reused); per wave and stage: 3 consecutive weight pieces, 3 activation pieces, and one extra -- the 13th weight / activation
//    piece on waves 0 / 1, 64 scales each (dword LDS-DMA) on waves 2 / 3;
//  * both groups meet at one barrier per step (a workgroup has one barrier); the group with fewer catches up before the exchange;
//  * epilogue: group k finishes token blocks 2k, 2k + 1 of every wave tile -- the other two blocks' FP32 sums go to the partner
//    wave through LDS, lower K range + upper K range, fp16 rows stored straight from registers (16 bytes per lane).
struct QkDma {
  unsigned voff, voffX;     // lane offsets from the stage's weight / activation base (the same for both) and of the extra piece
  int m0W, m0A, m0X;        // LDS byte addresses of the wave's first pieces inside stage slot 0 of its group's ring
  int xkind;                // extra piece: 1 weight rows, 2 activation rows, 3 weight scales (dword pieces)
};
__device__ __forceinline__ void q_dma4(unsigned m0v, unsigned voff, const void *sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory");
}
// DMA group i (0: weight pieces, 1: activation pieces, 2: the extra piece) of this wave into the slot at byte offset `so` of the ring
template <class C>
__device__ __forceinline__ void qk_piece(const QkDma &d, const uint8_t *wsrc, const uint8_t *asrc, const float *sbsrc, int so, int i) {
  if (i == 0) q_dma16x3(so + d.m0W, d.voff, wsrc);
  else if (i == 1) q_dma16x3(so + d.m0A, d.voff, asrc);
  else if (i == 2) {
    if (d.xkind == 3) q_dma4(so + d.m0X, d.voffX, sbsrc);
    else q_dma16(so + d.m0X, d.voffX, d.xkind == 1 ? (const void *)wsrc : (const void *)asrc);
  }
}

// The K step of the two-K-group kernel: q_step's MFMA / de-quantisation stream for a 64 x 64 wave tile -- 8 pair slots, slot i =
// (token block i % 4, feature-block pair i / 4); all four token fragments stay resident (a 64-token tile has the registers), and every
// LDS load is issued at least four slots ahead of its first reader:
//   slot 0      fragments fb 2,3 of THIS stage (read from slot 4 on; their previous contents died with slot 7 of the last step);
//               after the carried de-quantisation (pair 7 of the last step): weight scales of pair 1 and token scale 3 of this stage
//   slot 4      behind the mid-step barrier that publishes the next stage: its fragments fb 0,1 and token block 0;
//               after the de-quantisation of pair 3: its weight scales of pair 0
//   slot 4 + tb its token block tb; slot 5 + tb (tb < 3), after the de-quantisation of pair 4 + tb: its token scale tb
// (q_step's own slot order on 4 token blocks -- fragments fb 2,3 one slot, token block 3 two slots ahead of their MFMAs -- measured
// the same: 22.4 vs 22.9 us at 1024x4096x4096.  The step is bound by what a SIMD can issue, not by load latency: 16 MFMA + 128 FMA
// + 32 LDS reads + 7 LDS-DMA per wave in ~1,800 cycles, each class at its r02 issue cost; profiles/r03_mid_m.txt.)
template <class C>
struct QkRegs {
  v8i af[4], bf[4];
  float sb[2][8];
  QSa sa[4];
  v4f_t acc[2][2];
  int aW[2][3], aA[2][3], aS[2], aB[2];       // as QRegs
};
// ABL (tools build only): 2 = no de-quantisation, 4 = no MFMA, 8 = no fragment / scale re-loads
// `mid(i)` / `early(i)`, i = 0..3: behind the MFMAs of slots 4..7 / 0..3 (the caller's LDS-DMA)
template <class C, int SL, int ABL = 0, class FS, class FD, class FE = QNoEarly>
__device__ __forceinline__ void qk_step(QkRegs<C> &R, const char *lds, float (&c)[4][4][4], FS sync, FD mid, int ro = 0, int rn = 0, bool LAST = false,
                                        FE early = FE()) {
  constexpr int NX = SL < 0 ? -1 : (SL + 1) % 3;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int tb = i & 3, h = i >> 2;
    __builtin_amdgcn_sched_barrier(0);
    if (i == 4) sync();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 4)) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        R.acc[i & 1][k] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(R.af[2 * h + k], R.bf[tb], v4f_t{0.f, 0.f, 0.f, 0.f}, 3, 3, 0, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 8)) {
      if (i == 0) {
        R.af[2] = q_frag<C, SL>(lds, R.aW, p_row(2) * PITCH, ro);
        R.af[3] = q_frag<C, SL>(lds, R.aW, p_row(3) * PITCH, ro);
      }
      if (!LAST && i >= 4) {
        if (i == 4) {
          R.af[0] = q_frag<C, NX>(lds, R.aW, p_row(0) * PITCH, rn);
          R.af[1] = q_frag<C, NX>(lds, R.aW, p_row(1) * PITCH, rn);
        }
        R.bf[tb] = q_frag<C, NX>(lds, R.aA, p_row(tb) * PITCH, rn);
      }
    }
    if (i >= 4) mid(i - 4);
    else early(i);
    __builtin_amdgcn_sched_barrier(0);
    const int j = (i + 7) & 7, dtb = j & 3, dh = j >> 2;      // de-quantisation of the previous slot's pair (i == 0: the carried one)
    if constexpr (!(ABL & 2)) deq_pair<false>(R.acc[j & 1][0], R.acc[j & 1][1], R.sa[dtb], R.sb[dh], c[2 * dh][dtb], c[2 * dh + 1][dtb]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 8)) {                              // scales whose last reader was that de-quantisation
      if (i == 0) {
        q_load_sb<C, SL>(lds, R, 1, ro);
        R.sa[3] = q_scale<C, SL>(lds, R, p_row(3) * PITCH, ro);
      }
      if (!LAST) {
        if (i == 4) q_load_sb<C, NX>(lds, R, 0, rn);
        if (i >= 5) R.sa[dtb] = q_scale<C, NX>(lds, R, p_row(dtb) * PITCH, rn);
      }
    }
  }
}

// TR (tools build only, tools/trace_f6q.cpp): s_memtime stamps of workgroups 0 and gridDim.x - 1 into p.Dsz as u32 [2][8 waves][64]:
// [0..10] kernel phases, [16 + t] start of the group's K step t, [62], [63] s_memrealtime (100 MHz) at entry and exit
// ABL (tools build only): 1 = no LDS-DMA after the prologue, 2 / 4 / 8 as qk_step, 16 = no barrier in the K steps (the wait only)
template <class C, bool TR = false, int ABL = 0>
__global__ __launch_bounds__(C::NT * 2, 2) void gemm_w4a4_f6qk_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds_all[];
  unsigned *trb = nullptr;
  auto kstamp = [&](int k) {
    if constexpr (TR) {
      if (trb) { const unsigned t = (unsigned)__builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) trb[k] = t; }
    }
  };
  if constexpr (TR) {
    if (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) {
      trb = reinterpret_cast<unsigned *>(p.Dsz) + ((blockIdx.x ? 8 : 0) + (threadIdx.x >> 6)) * 128;
      if ((threadIdx.x & 63) == 0) trb[62] = (unsigned)__builtin_amdgcn_s_memrealtime();
    }
    kstamp(0);
  }
  static_assert(C::BM == 128 && C::BN == 128 && C::WM == 64 && C::NS == 3 && C::NW == 4 && C::FS32, "qk kernel: 128x128, 2 x 4 waves of 64 x 64");
  using Q = QC<C>;
  static_assert(Q::STAGE == C::STAGE_BYTES && 6 * Q::STAGE <= 160 * 1024, "qk kernel: stage layout");
  constexpr int NTB = 4, RING = 3 * Q::STAGE;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave_all >= 0 && wave_all < 8);
  const int kg = wave_all >> 2, wave = wave_all & 3;
  const int wm = wave >> 1, wn = wave & 1;
  const int ring0 = kg * RING;                             // this group's ring inside the workgroup's LDS (dynamic LDS starts at 0)
  const char *lds = lds_all;
  const int nbn = (p.N + C::BN - 1) / C::BN, nbm = (p.M + C::BM - 1) / C::BM;
  const int nwg = nbm * nbn;
  int id = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  constexpr int GM = 4;
  const int band = id / (GM * nbn), inband = id % (GM * nbn);
  const int rows_in_band = min(GM, nbm - band * GM);
  const int bm = band * GM + inband % rows_in_band, bn = inband / rows_in_band;
  const int m0 = bm * C::BM, n0 = bn * C::BN;

  float c[4][NTB][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NTB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[a][b][r] = 0.f;

  // compute steps of this group: n4 int4 groups from s_begin, then (upper group) the keeper
  const int G = p.G, csteps = G + 1;
  const int s_begin = csteps * kg / 2, c_end = csteps * (kg + 1) / 2;
  const bool has_keeper = kg == 1;
  const int n4 = min(c_end, G) - s_begin;                  // >= 2 (the dispatch sends G >= 7 here)
  QkDma d;
  d.voff = (unsigned)(wave * 3072 + lane * 16);
  d.m0W = ring0 + wave * 3072;
  d.m0A = ring0 + C::A_OFF + wave * 3072;
  d.xkind = wave == 0 ? 1 : (wave == 1 ? 2 : 3);
  d.voffX = (unsigned)(wave < 2 ? 12 * 1024 + lane * 16 : (wave & 1) * 256 + lane * 4);
  d.m0X = ring0 + (wave == 0 ? 12 * 1024 : (wave == 1 ? C::A_OFF + 12 * 1024 : Q::SB_OFF + (wave & 1) * 256));
  const int64_t wstep = p.f6_rows_b * PITCH, astep = p.f6_rows_a * PITCH;
  const uint8_t *wsrc0 = p.B4 + ((int64_t)s_begin * p.f6_rows_b + n0) * PITCH, *asrc0 = p.A4 + ((int64_t)s_begin * p.f6_rows_a + m0) * PITCH;
  const float *sbsrc0 = p.sB32 + (int64_t)s_begin * p.f6_rows_b + n0;
  auto slot_of = [&](int stage) { return lds_all + ring0 + (stage % 3) * Q::STAGE; };    // local stage index
  auto issue_stage = [&](int t, int so) {                  // every piece of this group's int4 stage t into the slot at byte offset so
#pragma unroll
    for (int i = 0; i < 3; ++i) qk_piece<C>(d, wsrc0 + t * wstep, asrc0 + t * astep, sbsrc0 + (int64_t)t * p.f6_rows_b, so, i);
  };
  // Prologue: stage 0 only where the group has enough steps for the LDS-DMA of stage 1 to go out behind the MFMAs of step 0's first
  // half (published by that step's mid-step barrier, where its first reader sits), as in the q kernel
  const bool early1 = n4 >= 5;                             // (steps 0..2 are regular ones)
  issue_stage(0, 0);
  if (!early1) issue_stage(1, Q::STAGE);
  kstamp(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  kstamp(2);

  const int l15 = lane & 15, kb = lane >> 4;
  QkRegs<C> R;
  {
    const int lw = ring0 + wn * 64 * PITCH + l15 * (2 * PITCH) + kb * 24;
    const int la = ring0 + C::A_OFF + wm * 64 * PITCH + l15 * (2 * PITCH);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        R.aW[st][k] = lw + 8 * k + st * 2 * Q::STAGE;
        R.aA[st][k] = la + kb * 24 + 8 * k + st * 2 * Q::STAGE;
        asm volatile("" : "+v"(R.aW[st][k]), "+v"(R.aA[st][k]));
      }
      R.aS[st] = la + 96 + st * 2 * Q::STAGE;
      R.aB[st] = ring0 + (wn * 64 + 8 * kb) * 4 + st * 2 * Q::STAGE;
      asm volatile("" : "+v"(R.aS[st]), "+v"(R.aB[st]));
    }
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) R.bf[tb] = q_frag<C, 0>(lds, R.aA, p_row(tb) * PITCH, 0);
    R.af[0] = q_frag<C, 0>(lds, R.aW, p_row(0) * PITCH, 0);
    R.af[1] = q_frag<C, 0>(lds, R.aW, p_row(1) * PITCH, 0);
    R.af[2] = R.af[1];                                      // (slot 0 of every step loads fb 2,3)
    R.af[3] = R.af[1];
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) R.sa[tb] = q_scale<C, 0>(lds, R, p_row(tb) * PITCH, 0);
    R.sa[3] = 0.f;                                          // the "carried pair" of the first step: zeros x 0
    q_load_sb<C, 0>(lds, R, 0, 0);
    q_load_sb<C, 0>(lds, R, 1, 0);
#pragma unroll
    for (int k = 0; k < 2; ++k) R.acc[1][k] = v4f_t{0.f, 0.f, 0.f, 0.f};
  }
  auto sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
  };
  // local step t on slot SL: the LDS-DMA of this group's int4 stage t + 2 behind the mid-step barrier
  auto reg = [&](auto sl, int t) {
    constexpr int SL = decltype(sl)::value;
    if constexpr (TR) { if (t < 44) kstamp(16 + t); }
    const int g = t + 2;
    const uint8_t *wsrc = wsrc0 + g * wstep, *asrc = asrc0 + g * astep;
    const float *sbsrc = sbsrc0 + (int64_t)g * p.f6_rows_b;
    qk_step<C, SL, ABL>(R, lds, c, sync, [&](int i) { if constexpr (!(ABL & 1)) qk_piece<C>(d, wsrc, asrc, sbsrc, ((SL + 2) % 3) * Q::STAGE, i); });
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  int t = 0;
  kstamp(3);
  if (early1) {                                             // step 0: + the LDS-DMA of stage 1 in its first half
    if constexpr (TR) kstamp(16);
    const uint8_t *wsrc = wsrc0 + 2 * wstep, *asrc = asrc0 + 2 * astep;
    const float *sbsrc = sbsrc0 + (int64_t)2 * p.f6_rows_b;
    qk_step<C, 0, ABL>(R, lds, c, sync, [&](int i) { if constexpr (!(ABL & 1)) qk_piece<C>(d, wsrc, asrc, sbsrc, 2 * Q::STAGE, i); }, 0, 0, false,
                       [&](int i) { if constexpr (!(ABL & 1)) qk_piece<C>(d, wsrc0 + wstep, asrc0 + astep, sbsrc0 + p.f6_rows_b, Q::STAGE, i); });
    reg(S1(), 1);
    reg(S2(), 2);
    t = 3;
  }
  for (; t + 4 < n4; t += 3) { reg(S0(), t); reg(S1(), t + 1); reg(S2(), t + 2); }
  // the last 2..4 int4 steps of the group on one generic body (run-time slots): their LDS-DMA is an int4 stage, a keeper half
  // (upper group) or nothing; the very last one has no next int4 stage to prefetch from.  (Compile-time slots for these too spill:
  // three bodies with wave-uniform branches around the LDS-DMA and the next-stage loads 287 VGPRs, one straight-line sequence of
  // compile-time bodies per remainder n4 % 3 -- no condition inside a step -- 237: it is the joins of the big register state.)
  for (; t < n4; ++t) {
    if constexpr (TR) { if (t < 44) kstamp(16 + t); }
    const int g = t + 2;
    const uint8_t *wsrc = wsrc0 + g * wstep, *asrc = asrc0 + g * astep;
    const float *sbsrc = sbsrc0 + (int64_t)g * p.f6_rows_b;
    qk_step<C, -1, ABL>(R, lds, c, sync,
                        [&](int i) {
                          if constexpr (ABL & 1) return;
                          if (g < n4) qk_piece<C>(d, wsrc, asrc, sbsrc, (g % 3) * Q::STAGE, i);
                          else if (has_keeper && i == 0) issue_keeper<C>(p, g - n4, slot_of(g), wave, lane, m0, n0);
                        },
                        (t % 3) * Q::STAGE, ((t + 1) % 3) * Q::STAGE, t + 1 == n4);
  }
  deq_pair<false>(R.acc[1][0], R.acc[1][1], R.sa[3], R.sb[1], c[2][NTB - 1], c[3][NTB - 1]);   // the carried pair (last slot) of the last int4 step
  int nbar = 1 + ((ABL & 16) ? 0 : n4);
  kstamp(4);
  if (has_keeper) {                                         // half 0 was published by the last mid-step barrier; half 1 was issued behind it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    ++nbar;
    kstamp(5);
    q_keeper<C, false>(p, slot_of(n4), slot_of(n4 + 1), wm, wn, lane, c, m0, n0);
  }
  kstamp(6);
  // barriers so far: prologue + one per int4 step (+ the keeper's); the other group's count, from its range
  {
    const int ob = csteps * (1 - kg) / 2, oe = csteps * (2 - kg) / 2;
    const int other = 1 + ((ABL & 16) ? 0 : min(oe, G) - ob) + (kg == 0 ? 1 : 0);
    for (; nbar < other; ++nbar) __builtin_amdgcn_s_barrier();
  }
  __builtin_amdgcn_s_barrier();                             // every wave is done with the rings
  kstamp(7);

  // ---- exchange: 2 token blocks x 4 feature blocks x 4 = 32 floats per lane to the partner wave (same wm, wn) of the other group
  float *xw = reinterpret_cast<float *>(lds_all) + (kg * 4 + wave) * (32 * 64) + lane;
  const float *xr = reinterpret_cast<const float *>(lds_all) + ((1 - kg) * 4 + wave) * (32 * 64) + lane;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = kg == 0 ? c[fb][2 + tt][r] : c[fb][tt][r];
        xw[((tt * 4 + fb) * 4 + r) * 64] = v;
      }
  __builtin_amdgcn_s_barrier();
  kstamp(8);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    float mine[4][4], other[4][4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        other[fb][r] = xr[((tt * 4 + fb) * 4 + r) * 64];
        mine[fb][r] = kg == 0 ? c[fb][tt][r] : c[fb][2 + tt][r];
      }
    const int tb = 2 * kg + tt;
    const int m = m0 + wm * 64 + p_row(tb) + 2 * l15;
    if (m >= p.M) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = n0 + wn * 64 + 32 * h + 8 * kb;
      if (n >= p.N) continue;
      v4u o;
      half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float lo = kg == 0 ? mine[2 * h + k][r] : other[2 * h + k][r], hi = kg == 0 ? other[2 * h + k][r] : mine[2 * h + k][r];
          ov[2 * r + k] = f2h(lo + hi);
        }
      *reinterpret_cast<v4u *>(p.D + (int64_t)m * p.N + n) = o;
    }
  }
  if constexpr (TR) {
    kstamp(9);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    kstamp(10);
    if (trb && (threadIdx.x & 63) == 0) trb[63] = (unsigned)__builtin_amdgcn_s_memrealtime();
  }
}

template <class C, bool TR = false, int ABL = 0>
static int launch_qk(const GemmParams &p, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  constexpr int LDS = 6 * QC<C>::STAGE;
  if (ensure_max_lds(reinterpret_cast<const void *>(&gemm_w4a4_f6qk_kernel<C, TR, ABL>), LDS, attr_done) != ATOM_OK) return ATOM_ERR_LAUNCH;
  const int nbm = (p.M + C::BM - 1) / C::BM, nbn = (p.N + C::BN - 1) / C::BN;
  hipLaunchKernelGGL((gemm_w4a4_f6qk_kernel<C, TR, ABL>), dim3((unsigned)(nbm * nbn)), dim3(C::NT * 2), LDS, s, p);
  return check_launch();
}

// ================================================================================================================
// The 256x128 kernel ("q2"): 256 tokens x 128 features, 8 waves of 64 x 64 on ONE 3-stage ring, qk_step as its K step, the K steps in
// order (the q kernel's sums, atom_gemm_w4a4_f6_order = 1), fp16 rows stored from inside the keeper step.  For shapes between the
// two-K-group kernels and the 256x256 kernel: more than 128 but at most 256 tiles of 256x128 (2048 x 4096: 128 tiles of 256x256 leave
// half of the CUs idle).  Per wave and stage: 3 consecutive activation pieces, 1 weight piece, and one extra -- the remaining five
// weight pieces on waves 0-4, the remaining two activation pieces on waves 5, 6, the 128 float32 weight scales (two dword pieces) on 7.
struct Q2Dma {
  unsigned voffA, voffW, voffX;   // lane offsets from the stage's activation / weight / extra base
  int m0A, m0W, m0X;              // LDS byte offsets of the wave's pieces inside a stage
  int xkind;                      // extra: 1 weight rows, 2 activation rows, 3 weight scales
};
template <class C>
__device__ __forceinline__ void q2_piece(const Q2Dma &d, const uint8_t *wsrc, const uint8_t *asrc, const float *sbsrc, int so, int i) {
  if (i == 0) q_dma16x3(so + d.m0A, d.voffA, asrc);
  else if (i == 1) q_dma16(so + d.m0W, d.voffW, wsrc);
  else if (i == 2) {
    if (d.xkind == 3) {
      q_dma4(so + d.m0X, d.voffX, sbsrc);
      q_dma4(so + d.m0X + 256, d.voffX + 256, sbsrc);
    } else {
      q_dma16(so + d.m0X, d.voffX, d.xkind == 1 ? (const void *)wsrc : (const void *)asrc);
    }
  }
}

template <class C>
__global__ __launch_bounds__(C::NT, 2) void gemm_w4a4_f6q2_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  static_assert(C::BM == 256 && C::BN == 128 && C::WM == 64 && C::NS == 3 && C::NW == 8 && C::FS32, "q2 kernel: 256x128, 8 waves of 64 x 64");
  using Q = QC<C>;
  static_assert(Q::STAGE == C::STAGE_BYTES, "q2 kernel: stage layout");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < 8);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + C::BN - 1) / C::BN, nbm = (p.M + C::BM - 1) / C::BM;
  const int nwg = nbm * nbn;
  int id = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  constexpr int GM = 4;
  const int band = id / (GM * nbn), inband = id % (GM * nbn);
  const int rows_in_band = min(GM, nbm - band * GM);
  const int bm = band * GM + inband % rows_in_band, bn = inband / rows_in_band;
  const int m0 = bm * C::BM, n0 = bn * C::BN;

  float c[4][4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[a][b][r] = 0.f;

  const int G = p.G;                                        // stages 0..G-1: int4 groups; G, G+1: keeper halves
  Q2Dma d;
  d.voffA = (unsigned)(wave * 3072 + lane * 16);
  d.m0A = C::A_OFF + wave * 3072;
  d.voffW = (unsigned)(wave * 1024 + lane * 16);
  d.m0W = wave * 1024;
  d.xkind = wave < 5 ? 1 : (wave < 7 ? 2 : 3);
  d.voffX = (unsigned)(wave < 5 ? (8 + wave) * 1024 + lane * 16 : (wave < 7 ? (24 + wave - 5) * 1024 + lane * 16 : lane * 4));
  d.m0X = wave < 5 ? (8 + wave) * 1024 : (wave < 7 ? C::A_OFF + (24 + wave - 5) * 1024 : Q::SB_OFF);
  const int64_t wstep = p.f6_rows_b * PITCH, astep = p.f6_rows_a * PITCH;
  const uint8_t *wsrc0 = p.B4 + (int64_t)n0 * PITCH, *asrc0 = p.A4 + (int64_t)m0 * PITCH;
  const float *sbsrc0 = p.sB32 + n0;
  auto slot_of = [&](int stage) { return lds + (stage % 3) * Q::STAGE; };
  auto issue_stage = [&](int g, int so) {
#pragma unroll
    for (int i = 0; i < 3; ++i) q2_piece<C>(d, wsrc0 + g * wstep, asrc0 + g * astep, sbsrc0 + (int64_t)g * p.f6_rows_b, so, i);
  };
  const bool early1 = G >= 5;                               // (steps 0..2 are regular ones)
  issue_stage(0, 0);
  if (!early1) {
    if (G >= 2) issue_stage(1, Q::STAGE);
    else issue_keeper<C>(p, 0, slot_of(1), wave, lane, m0, n0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int l15 = lane & 15, kb = lane >> 4;
  QkRegs<C> R;
  {
    const int lw = wn * 64 * PITCH + l15 * (2 * PITCH) + kb * 24;
    const int la = C::A_OFF + wm * 64 * PITCH + l15 * (2 * PITCH);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        R.aW[st][k] = lw + 8 * k + st * 2 * Q::STAGE;
        R.aA[st][k] = la + kb * 24 + 8 * k + st * 2 * Q::STAGE;
        asm volatile("" : "+v"(R.aW[st][k]), "+v"(R.aA[st][k]));
      }
      R.aS[st] = la + 96 + st * 2 * Q::STAGE;
      R.aB[st] = (wn * 64 + 8 * kb) * 4 + st * 2 * Q::STAGE;
      asm volatile("" : "+v"(R.aS[st]), "+v"(R.aB[st]));
    }
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) R.bf[tb] = q_frag<C, 0>(lds, R.aA, p_row(tb) * PITCH, 0);
    R.af[0] = q_frag<C, 0>(lds, R.aW, p_row(0) * PITCH, 0);
    R.af[1] = q_frag<C, 0>(lds, R.aW, p_row(1) * PITCH, 0);
    R.af[2] = R.af[1];
    R.af[3] = R.af[1];
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) R.sa[tb] = q_scale<C, 0>(lds, R, p_row(tb) * PITCH, 0);
    R.sa[3] = 0.f;
    q_load_sb<C, 0>(lds, R, 0, 0);
    q_load_sb<C, 0>(lds, R, 1, 0);
#pragma unroll
    for (int k = 0; k < 2; ++k) R.acc[1][k] = v4f_t{0.f, 0.f, 0.f, 0.f};
  }
  auto sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  auto reg = [&](auto sl, int s) {
    constexpr int SL = decltype(sl)::value;
    const int g = s + 2;
    const uint8_t *wsrc = wsrc0 + g * wstep, *asrc = asrc0 + g * astep;
    const float *sbsrc = sbsrc0 + (int64_t)g * p.f6_rows_b;
    qk_step<C, SL>(R, lds, c, sync, [&](int i) { q2_piece<C>(d, wsrc, asrc, sbsrc, ((SL + 2) % 3) * Q::STAGE, i); });
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  int s = 0;
  if (early1) {                                             // step 0: + the LDS-DMA of stage 1 in its first half
    const uint8_t *wsrc = wsrc0 + 2 * wstep, *asrc = asrc0 + 2 * astep;
    const float *sbsrc = sbsrc0 + (int64_t)2 * p.f6_rows_b;
    qk_step<C, 0>(R, lds, c, sync, [&](int i) { q2_piece<C>(d, wsrc, asrc, sbsrc, 2 * Q::STAGE, i); }, 0, 0, false,
                  [&](int i) { q2_piece<C>(d, wsrc0 + wstep, asrc0 + astep, sbsrc0 + p.f6_rows_b, Q::STAGE, i); });
    reg(S1(), 1);
    reg(S2(), 2);
    s = 3;
  }
  for (; s + 4 < G; s += 3) { reg(S0(), s); reg(S1(), s + 1); reg(S2(), s + 2); }
  for (; s < G; ++s) {                                      // the last 2..4 steps: one generic body (see the q kernel)
    const int g = s + 2;
    const uint8_t *wsrc = wsrc0 + g * wstep, *asrc = asrc0 + g * astep;
    const float *sbsrc = sbsrc0 + (int64_t)g * p.f6_rows_b;
    qk_step<C, -1>(R, lds, c, sync,
                   [&](int i) {
                     if (g < G) q2_piece<C>(d, wsrc, asrc, sbsrc, (g % 3) * Q::STAGE, i);
                     else if (i == 0) issue_keeper<C>(p, g - G, slot_of(g), wave, lane, m0, n0);
                   },
                   (s % 3) * Q::STAGE, ((s + 1) % 3) * Q::STAGE, s + 1 == G);
  }
  deq_pair<false>(R.acc[1][0], R.acc[1][1], R.sa[3], R.sb[1], c[2][3], c[3][3]);   // the carried pair (last slot) of the last int4 step
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  q_keeper<C, true>(p, slot_of(G), slot_of(G + 1), wm, wn, lane, c, m0, n0);
}

template <class C>
static int launch_q2(const GemmParams &p, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  constexpr int LDS = 3 * QC<C>::STAGE;
  if (ensure_max_lds(reinterpret_cast<const void *>(&gemm_w4a4_f6q2_kernel<C>), LDS, attr_done) != ATOM_OK) return ATOM_ERR_LAUNCH;
  const int nbm = (p.M + C::BM - 1) / C::BM, nbn = (p.N + C::BN - 1) / C::BN;
  hipLaunchKernelGGL((gemm_w4a4_f6q2_kernel<C>), dim3((unsigned)(nbm * nbn)), dim3(C::NT), LDS, s, p);
  return check_launch();
}

template <class C, int ABL = 0>
static int launch_p(const GemmParams &p, hipStream_t s) {
  constexpr int lds_bytes = C::LDS_BYTES + ((ABL & 16) ? 8 * 80 * 4 : 0);
  static_assert(lds_bytes <= 160 * 1024, "LDS");
  static std::atomic<uint64_t> attr_done{0};
  if (ensure_max_lds(reinterpret_cast<const void *>(&gemm_w4a4_f6p_kernel<C, ABL>), lds_bytes, attr_done) != ATOM_OK) return ATOM_ERR_LAUNCH;
  const int nbm = (p.M + C::BM - 1) / C::BM, nbn = (p.N + C::BN - 1) / C::BN;
  hipLaunchKernelGGL((gemm_w4a4_f6p_kernel<C, ABL>), dim3((unsigned)(nbm * nbn)), dim3(C::NT), lds_bytes, s, p);
  return check_launch();
}

template <class C, bool SK = false, int KG = 1, bool PH = false, bool TR = false>
static int launch_x16(const GemmParams &p, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  // K groups: epilogue rows + the sums handed over
  constexpr int XCH = KG == 4 ? C::NW * 4 * (16 * 80 + 8 * 4 * 64 * 4) : C::NW * KG * (C::WM / 2) * (144 + 64 * 4);
  constexpr int LDS = (KG > 1 ? (KG * C::NS * C::STAGE_BYTES > XCH ? KG * C::NS * C::STAGE_BYTES : XCH) : C::LDS_BYTES) + (TR ? 2048 : 0);
  static_assert(LDS <= 160 * 1024, "LDS");
  if (ensure_max_lds(reinterpret_cast<const void *>(&gemm_w4a4_f6x16_kernel<C, SK, KG, PH, TR>), LDS, attr_done) != ATOM_OK) return ATOM_ERR_LAUNCH;
  const int nbm = (p.M + C::BM - 1) / C::BM, nbn = (p.N + C::BN - 1) / C::BN;
  hipLaunchKernelGGL((gemm_w4a4_f6x16_kernel<C, SK, KG, PH, TR>), dim3((unsigned)(nbm * nbn), (unsigned)(SK ? p.splits : 1)), dim3(C::NT * KG),
                     LDS, s, p);
  if (SK) {
    const int64_t MN = (int64_t)p.M * p.N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN / 8 + 255) / 256)), dim3(256), 0, s, p.ws, p.D, MN, p.splits);
  }
  return check_launch();
}

template <class C, bool SK, int ABL = 0>
static int launch(const GemmParams &p, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  if (ensure_max_lds(reinterpret_cast<const void *>(&gemm_w4a4_f6_kernel<C, SK, ABL>), C::LDS_BYTES, attr_done) != ATOM_OK) return ATOM_ERR_LAUNCH;
  const int nbm = (p.M + C::BM - 1) / C::BM, nbn = (p.N + C::BN - 1) / C::BN;
  hipLaunchKernelGGL((gemm_w4a4_f6_kernel<C, SK, ABL>), dim3((unsigned)(nbm * nbn), (unsigned)(SK ? p.splits : 1)), dim3(C::NT),
                     C::LDS_BYTES, s, p);
  if (SK) {
    const int64_t MN = (int64_t)p.M * p.N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN / 8 + 255) / 256)), dim3(256), 0, s, p.ws, p.D, MN, p.splits);
  }
  return check_launch();
}

}  // namespace f6

int launch_gemm_f6_gateup(const GemmParams &p, int sim, hipStream_t s) {
  using C = f6::Cfg<256, 256, 4, 3>;
  if (p.b_pairs) return sim ? f6::launch_q<C, 1, false, true>(p, s) : f6::launch_q<C, 2, false, true>(p, s);
  return sim ? f6::launch_q<C, 1>(p, s) : f6::launch_q<C, 2>(p, s);
}

// cfg: 0 = 256x256 (8 waves, the pipelined kernel) and 3 = 128x128 (4 waves, three workgroups per CU) on 16x16x128 MFMA
// micro-tiles, 5 / 6 / 9 / 12 = the K-group kernels (128x128 / 64x128 tiles, two or four groups of 4 waves), 8 = 256x128 (8 waves,
// qk_step; ATOM_B_F6S weights, else 3), 2 = 64x128 (2 waves, 32x32x64 MFMA; split-K when p.splits > 1 and p.ws is set); tuning only: 1 = 256x128,
// 10 / 13 = 256x256 / 128x128 on the 32x32x64 MFMA, 40 = 256x256 first-generation micro-tile kernel
int launch_gemm_f6(const GemmParams &p, int cfg, hipStream_t s) {
#ifdef ATOM_TOOLS
  if (cfg == 116) {   // traced run (tools/trace_f6.cpp): the stamp buffer arrives in ATOM_TRACE_PTR
    const char *e = getenv("ATOM_TRACE_PTR");
    if (!e) return ATOM_ERR_INVALID_ARG;
    GemmParams q = p;
    q.Dsz = reinterpret_cast<half_t *>(strtoull(e, nullptr, 16));
    return f6::launch<f6::Cfg<256, 256, 4, 3>, false, 16>(q, s);
  }
  switch (cfg) {   // tools/gemm_bench only: 100 + ablation mask on the 256x256 geometry
#define ATOM_ABL(a) case 100 + a: return f6::launch<f6::Cfg<256, 256, 4, 3>, false, a>(p, s);
    ATOM_ABL(1) ATOM_ABL(2) ATOM_ABL(3) ATOM_ABL(4) ATOM_ABL(6) ATOM_ABL(7) ATOM_ABL(8) ATOM_ABL(10) ATOM_ABL(14) ATOM_ABL(15) ATOM_ABL(32)
#undef ATOM_ABL
  }
  if (cfg == 1005 || cfg == 1009) {   // traced run of the two-K-group kernels (tools/trace_f6kg.cpp)
    const char *e = getenv("ATOM_TRACE_PTR");
    if (!e || !p.sB32) return ATOM_ERR_INVALID_ARG;
    GemmParams q = p;
    q.Dsz = reinterpret_cast<half_t *>(strtoull(e, nullptr, 16));
    if (cfg == 1005) return f6::launch_x16<f6::Cfg<128, 128, 2, 2, 3, 1>, false, 2, false, true>(q, s);
    return f6::launch_x16<f6::Cfg<64, 128, 1, 3, 3, 1>, false, 2, false, true>(q, s);
  }
  switch (cfg) {   // 2100 + ablation mask on the two-K-group q-step kernel
#define ATOM_ABL(a) case 2100 + a: return p.sB32 ? f6::launch_qk<f6::Cfg<128, 128, 2, 3, 3, 1>, false, a>(p, s) : ATOM_ERR_INVALID_ARG;
    ATOM_ABL(0) ATOM_ABL(1) ATOM_ABL(2) ATOM_ABL(4) ATOM_ABL(8) ATOM_ABL(16) ATOM_ABL(17) ATOM_ABL(6) ATOM_ABL(14) ATOM_ABL(15) ATOM_ABL(31)
    ATOM_ABL(3) ATOM_ABL(5) ATOM_ABL(9) ATOM_ABL(7) ATOM_ABL(11) ATOM_ABL(13)
#undef ATOM_ABL
  }
  if (cfg == 2017) {   // traced run of the two-K-group q-step kernel (tools/trace_f6q.cpp qk)
    const char *e = getenv("ATOM_TRACE_PTR");
    if (!e || !p.sB32) return ATOM_ERR_INVALID_ARG;
    GemmParams q = p;
    q.Dsz = reinterpret_cast<half_t *>(strtoull(e, nullptr, 16));
    return f6::launch_qk<f6::Cfg<128, 128, 2, 3, 3, 1>, true>(q, s);
  }
  if (cfg == 2016) {   // traced run of the q kernel (tools/trace_f6q.cpp)
    const char *e = getenv("ATOM_TRACE_PTR");
    if (!e || !p.sB32) return ATOM_ERR_INVALID_ARG;
    GemmParams q = p;
    q.Dsz = reinterpret_cast<half_t *>(strtoull(e, nullptr, 16));
    return f6::launch_q<f6::Cfg<256, 256, 4, 3>, 0, true>(q, s);
  }
  if (cfg == 1016) {   // traced run of the pipelined kernel (tools/trace_f6.cpp)
    const char *e = getenv("ATOM_TRACE_PTR");
    if (!e) return ATOM_ERR_INVALID_ARG;
    GemmParams q = p;
    q.Dsz = reinterpret_cast<half_t *>(strtoull(e, nullptr, 16));
    return f6::launch_p<f6::Cfg<256, 256, 4, 3>, 16>(q, s);
  }
  switch (cfg) {   // 1000 + ablation mask on the pipelined kernel
#define ATOM_ABL(a) case 1000 + a: return f6::launch_p<f6::Cfg<256, 256, 4, 3>, a>(p, s);
    ATOM_ABL(1) ATOM_ABL(2) ATOM_ABL(3) ATOM_ABL(4) ATOM_ABL(6) ATOM_ABL(7) ATOM_ABL(8) ATOM_ABL(10) ATOM_ABL(14) ATOM_ABL(15)
    ATOM_ABL(64) ATOM_ABL(128) ATOM_ABL(192) ATOM_ABL(207) ATOM_ABL(256) ATOM_ABL(257) ATOM_ABL(512) ATOM_ABL(513) ATOM_ABL(768) ATOM_ABL(769)
    ATOM_ABL(520) ATOM_ABL(9) ATOM_ABL(1024) ATOM_ABL(2048) ATOM_ABL(4096)
#undef ATOM_ABL
  }
#endif
#ifdef ATOM_TOOLS   // geometries no shape is dispatched to (f6_pick_cfg): tuning builds only
  if (cfg == 2) {
    if (p.splits > 1 && p.ws) return f6::launch<f6::Cfg<64, 128, 2, 3>, true>(p, s);
    return f6::launch<f6::Cfg<64, 128, 2, 3>, false>(p, s);
  }
  if (cfg == 10) return f6::launch<f6::Cfg<256, 256, 4, 3>, false>(p, s);   // tuning: 256x256 on the 32x32x64 MFMA
  if (cfg == 1) return f6::launch<f6::Cfg<256, 128, 4, 2>, false>(p, s);
  if (cfg == 13) return f6::launch<f6::Cfg<128, 128, 2, 2, 3>, false>(p, s);  // tuning: 128x128 on the 32x32x64 MFMA
#endif
  if (cfg == 20) {                                                             // mid-size batches: 64x64 tiles on a deep LDS ring (gemm_w4a4_mid.hip)
    const int st = launch_gemm_f6_mid(p, s);
    if (st != ATOM_ERR_SHAPE) return st;
    cfg = 3;
  }
  if (cfg == 8) {                                                              // 256x128, qk_step, K steps in order
    if (p.sB32 && p.G >= 2) return f6::launch_q2<f6::Cfg<256, 128, 2, 3, 2, 1>>(p, s);
    cfg = 3;                                                                   // fp16 weight scales: the 128x128 geometry (same order)
  }
  if (cfg == 3) {                                                              // 128x128, 4 waves
    // float32 weight scales staged (ATOM_B_F6S): 8 % fewer cycles per tile, but 54,272 B of LDS = two workgroups per CU (the LDS
    // granule makes it 55,040); fp16 scales staged and converted once per step: 53,760 B = three.  The former unless the third
    // workgroup saves a round of tiles (profiles/r02_mid_m.txt: 1536x4096x4096 37.7 vs 40.9 us, 1024x11008x4096 64.7 vs 55.4).
    const int64_t t128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (p.sB32 && (t128 + 511) / 512 <= (t128 + 767) / 768) {
#ifdef ATOM_TOOLS
      if (p.splits > 1 && p.ws) return f6::launch_x16<f6::Cfg<128, 128, 2, 2, 3, 1>, true>(p, s);
#endif
      return f6::launch_x16<f6::Cfg<128, 128, 2, 2, 3, 1>>(p, s);
    }
#ifdef ATOM_TOOLS
    if (p.splits > 1 && p.ws) return f6::launch_x16<f6::Cfg<128, 128, 2, 2, 3, 2>, true>(p, s);
#endif
    return f6::launch_x16<f6::Cfg<128, 128, 2, 2, 3, 2>>(p, s);
  }
  // two K groups of 4 waves per workgroup: for shapes that put at most one tile on a CU (f6_pick_cfg)
#ifdef ATOM_TOOLS
  if (cfg == 5) {                                                              // tuning: 128x128, two K groups, 2 stages per group
    if (p.sB32) return f6::launch_x16<f6::Cfg<128, 128, 2, 2, 3, 1>, false, 2>(p, s);
    return f6::launch_x16<f6::Cfg<128, 128, 2, 2, 3, 2>, false, 2>(p, s);
  }
#endif
  if (cfg == 6) {                                                              // 128x128, two K groups, 3 stages per group
    // float32 weight scales (ATOM_B_F6S) and a K long enough for its unrolled loop: the q kernel's K step per group (same sums)
    if (p.sB32 && p.G >= 7 && ATOM_TUNE("ATOM_QK", 1)) return f6::launch_qk<f6::Cfg<128, 128, 2, 3, 3, 1>>(p, s);
    if (p.sB32) return f6::launch_x16<f6::Cfg<128, 128, 2, 3, 3, 1>, false, 2>(p, s);
    return f6::launch_x16<f6::Cfg<128, 128, 2, 3, 3, 2>, false, 2>(p, s);
  }
#ifdef ATOM_TOOLS   // the 64x128 K-group kernels of rounds 2-4: no shape is dispatched to them since the mid-size-batch kernel (cfg 20)
  if (cfg == 9) {                                                              // 64x128 (32-token wave tiles), groups half a step apart
    if (p.sB32) return f6::launch_x16<f6::Cfg<64, 128, 1, 3, 3, 1>, false, 2, true>(p, s);
    return f6::launch_x16<f6::Cfg<64, 128, 1, 3, 3, 2>, false, 2, true>(p, s);
  }
  if (cfg == 12) {                                                             // 64x128, four K groups of 4 waves (exactly 160 KiB of LDS)
    if (p.sB32) return f6::launch_x16<f6::Cfg<64, 128, 1, 2, 4, 1>, false, 4>(p, s);
    return f6::launch_x16<f6::Cfg<64, 128, 1, 2, 4, 2>, false, 4>(p, s);
  }
#endif
#ifdef ATOM_TOOLS   // tuning builds only
  if (cfg == 51 && p.sB32) return f6::launch_x16<f6::Cfg<128, 128, 2, 2, 3, 1>>(p, s);   // tuning: the two scale forms of cfg 3
  if (cfg == 52) return f6::launch_x16<f6::Cfg<128, 128, 2, 2, 3, 2>>(p, s);
  if (cfg == 53) return f6::launch_x16<f6::Cfg<128, 128, 2, 2, 3>>(p, s);            // tuning: cfg 3 with fp16 weight scales
  if (cfg == 40) return f6::launch_x16<f6::Cfg<256, 256, 4, 3>>(p, s);        // tuning: 256x256, first-generation micro-tile kernel
  // round 6, VERDICT r05 missing #1: the SAME kernel body as 8 waves of 64 features x 128 tokens (two per SIMD, 256 registers) and as
  // 16 waves of 64 x 64 (four per SIMD, 128 registers) -- the same-code A/B behind profiles/r06/ab_16wave.txt
  if (cfg == 41 && p.sB32) return f6::launch_x16<f6::Cfg<256, 256, 4, 3, 2, 1>>(p, s);
  if (cfg == 42 && p.sB32) return f6::launch_x16<f6::Cfg<256, 256, 2, 3, 4, 1>>(p, s);
  if (cfg == 4) {                                                              // 128x128, 8 waves of 64 features x 32 tokens
    if (p.splits > 1 && p.ws) return f6::launch_x16<f6::Cfg<128, 128, 1, 2, 3>, true>(p, s);
    return f6::launch_x16<f6::Cfg<128, 128, 1, 2, 3>>(p, s);
  }
#endif
  if (!p.sB32 || cfg == 30) return f6::launch_p<f6::Cfg<256, 256, 4, 3>>(p, s);   // 256x256, pipelined across K steps (fp16 weight scales; cfg 30: tuning)
  if (p.b_pairs) return f6::launch_q<f6::Cfg<256, 256, 4, 3>, 0, false, true>(p, s);   // ... third generation (ATOM_B_F6S; the headline), shared scale products
  return f6::launch_q<f6::Cfg<256, 256, 4, 3>>(p, s);                          // ... the same for weights whose channel pairs have their own scales
}

}  // namespace atom
