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
//                              fields, byte 96..97 = the fp16 scale of (row, group), 98..103 zero
// Group-major makes the 256 rows a tile needs for one K step ONE contiguous 26 KiB block: the LDS-DMA is base + 16*lane
// with no per-row address, rows_pad (a multiple of 256) keeps tail tiles in bounds, and the 104-byte pitch (26 dwords =
// 2 x odd) makes the 32 rows of an MFMA fragment read (3 x ds_read_b64 of 24 bytes) hit 32 distinct bank pairs with no
// swizzle.  The token scale is read from the row itself (no scale DMA, no scale region); weight scales keep their dense
// fp16 array.  The INT8 keeper runs as the two 64-column half-steps of the INT8 kernel in the same stage buffer.
// Arithmetic is the same contract: t = round_f32(idot * sA), c = fma(t, sB, c) per group in order, keeper last --
// results are bit-identical to the INT8 kernels.
#include <cstdlib>
#include "common.h"

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
template <int BM_, int BN_, int TM_, int NS_, int OCC_ = 2>
struct Cfg {
  static constexpr int OCC = OCC_;                                     // waves per SIMD the register budget is set for
  static constexpr int BM = BM_, BN = BN_, TM = TM_, NS = NS_;
  static constexpr int WM = 32 * TM, WGM = BM / WM, WGN = BN / 64, NW = WGM * WGN, NT = NW * 64;
  static constexpr int W_BYTES = BN * PITCH, A_BYTES = BM * PITCH;
  static constexpr int NBW = W_BYTES / 1024;                           // whole 1 KiB DMA blocks of the weight rows
  static constexpr int NBA = (A_BYTES + 1023) / 1024;                  // activation rows: the last block may be partial
  static constexpr int A_TAIL = (A_BYTES % 1024) / 16;                 // lanes of that partial block (0 = it is whole)
  static constexpr int NSB = BN / 128;                                 // weight-scale pieces (128 fp16 = 64 dwords each)
  static constexpr int A_OFF = W_BYTES;
  static constexpr int SB_OFF = W_BYTES + NBA * 1024;                  // BN fp16 weight scales (int4 steps and keeper)
  static constexpr int STAGE_BYTES = SB_OFF + BN * 2;
  static constexpr int KP_SA_OFF = (BN + BM) * 64;                     // keeper half-steps: rows of 64 B, then BM dwords sA8
  static constexpr int NPIECE = NBW + NBA + NSB;                       // DMA instructions per int4 stage
  static constexpr int NKP = (BN + BM) / 16 + BM / 64 + NSB;           // ... per keeper half-step
  // The LDS-DMA instructions are issued by the first NDW waves.
  static constexpr int NDW = NW;   // (measured: NW / 2 -- only the older wave of each SIMD issuing -- is 6 % slower)
  static constexpr int GLDS = ((NPIECE > NKP ? NPIECE : NKP) + NDW - 1) / NDW;  // per issuing wave, padded with repeats
  static constexpr int EP_BYTES = NW * 64 * 144;
  static constexpr int LDS_BYTES = NS * STAGE_BYTES > EP_BYTES ? NS * STAGE_BYTES : EP_BYTES;
  static_assert(BN % 128 == 0 && BM % 64 == 0 && BM % WM == 0 && (TM == 2 || TM == 4) && NS >= 2, "geometry");
  static_assert(KP_SA_OFF + BM * 4 <= SB_OFF && LDS_BYTES <= 160 * 1024, "stage layout");
};

// DMA instruction i (of C::GLDS) of this wave for int4 group g
template <class C>
__device__ __forceinline__ void issue_int4_piece(const GemmParams &p, int g, char *slot, int wave, int lane, int m0, int n0, int i) {
  const uint8_t *wsrc = p.B4 + ((int64_t)g * p.f6_rows_b + n0) * PITCH;
  const uint8_t *asrc = p.A4 + ((int64_t)g * p.f6_rows_a + m0) * PITCH - C::W_BYTES;   // block j >= NBW is asrc + j*1024
  int j = i * C::NDW + wave;                               // piece j: NBW weight blocks, NBA activation blocks, NSB scales
  j = j < C::NPIECE ? j : j - C::NSB;                      // padding repeats a scale piece (same bytes, same place)
  if (i * C::NDW + C::NDW <= C::NBW + C::NBA - (C::A_TAIL ? 1 : 0)) {              // compile time: whole data blocks only
    const uint8_t *base = (i * C::NDW + C::NDW <= C::NBW || j < C::NBW) ? wsrc : asrc;
    __builtin_amdgcn_global_load_lds((gptr_t)(base + j * 1024 + lane * 16), (lptr_t)(slot + j * 1024), 16, 0, 0);
  } else if (j < C::NBW + C::NBA) {
    const uint8_t *base = j < C::NBW ? wsrc : asrc;
    // the partial last activation block runs with fewer lanes enabled (one instruction either way: vmcnt stays uniform)
    if (!C::A_TAIL || j < C::NBW + C::NBA - 1 || lane < C::A_TAIL)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + j * 1024 + lane * 16), (lptr_t)(slot + j * 1024), 16, 0, 0);
  } else {                                                 // 128 weight scales per piece, a dword (2 channels) per lane
    const int part = j - (C::NBW + C::NBA);
    const half_t *sBb = p.sB + (int64_t)g * p.N;
    const int n = min(n0 + part * 128 + 2 * lane, p.N - 2);
    __builtin_amdgcn_global_load_lds((gptr_t)(sBb + n), (lptr_t)(slot + C::SB_OFF + part * 256), 4, 0, 0);
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
#pragma unroll
  for (int i = 0; i < C::GLDS; ++i) {
    int j = i * C::NDW + wave;
    j = j < C::NKP ? j : ND + NSA + (j % C::NSB);          // padding repeats a weight-scale piece
    if (i * C::NDW + C::NDW <= ND || j < ND) {               // (first half: known at compile time)
      const int row = j * 16 + (lane >> 2);
      const unsigned idx = (unsigned)(j < C::BN / 16 ? min(n0 + row, p.N - 1) : min(m0 + row - C::BN, p.M - 1));
      const uint8_t *base = (j < C::BN / 16 ? p.B8 : p.A8) + half * 64;
      __builtin_amdgcn_global_load_lds((gptr_t)(base + idx * kKeeper + kj), (lptr_t)(slot + j * 1024), 16, 0, 0);
    } else if (j < ND + NSA) {                             // sA8 of 64 tokens: one fp16 per lane -> zero-extended dword
      const int pc = j - ND;
      const int idx = min(m0 + pc * 64 + lane, p.M - 1);
      const unsigned ksa = (unsigned)(p.ref_layout ? ref_scale_index(idx) : idx);
      __builtin_amdgcn_global_load_lds((gptr_t)(p.sA8 + ksa), (lptr_t)(slot + C::KP_SA_OFF + pc * 256), 2, 0, 0);
    } else {
      const int part = j - ND - NSA;
      const int n = min(n0 + part * 128 + 2 * lane, p.N - 2);
      __builtin_amdgcn_global_load_lds((gptr_t)(p.sB8 + n), (lptr_t)(slot + C::SB_OFF + part * 256), 4, 0, 0);
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
  // t = round_f32(idot * sA) for all 16 elements FIRST (in place, one block): a multiply followed directly by the FMA that
  // reads it costs 3.2 cycles per instruction instead of 2 (dependent issue), which was 9 us of the 4096^3 launch
#define M1(i) "v_mul_f32 %" #i ", %" #i ", %16\n"
  // element 0 through the compiler: its hazard recogniser then places the MFMA->VALU wait states the asm block relies on
  acc[0] *= sa;
  asm volatile("" : "+v"(acc[0]));
  asm volatile(M1(1) M1(2) M1(3) M1(4) M1(5) M1(6) M1(7) M1(8) M1(9) M1(10) M1(11) M1(12) M1(13) M1(14) M1(15)
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                 "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15])
               : "v"(sa));
#undef M1
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const half_t *hv = reinterpret_cast<const half_t *>(&sbp[r >> 2]);
    c[r] = __builtin_fmaf(acc[r], (float)hv[r & 3], c[r]);
    asm volatile("" : "+v"(c[r]));
  }
}

// one int4 group out of LDS: 2 BF6 MFMAs per 32x32 tile
// ABL (tools only, -DATOM_F6_ABLATE): 2 = no de-quantisation, 4 = no MFMA, 8 = no fragment refills, 16 = s_memtime
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

// keeper half-step out of LDS (INT8 MFMA, magic-biased accumulator), de-quantised per half
template <class C>
__device__ __forceinline__ void compute_keeper(const char *slot, int wm, int wn, int lane, float (&c)[TN][C::TM][16]) {
  constexpr int TM = C::TM;
  const int l31 = lane & 31, h = lane >> 5;
  const int sw = (l31 >> 2) & 3;
  const char *pw0 = slot + (wn * 64 + l31) * 64 + (((0 + h) ^ sw) << 4);
  const char *pw1 = slot + (wn * 64 + l31) * 64 + (((2 + h) ^ sw) << 4);
  const char *pa0 = slot + (C::BN + wm * C::WM + l31) * 64 + (((0 + h) ^ sw) << 4);
  const char *pa1 = slot + (C::BN + wm * C::WM + l31) * 64 + (((2 + h) ^ sw) << 4);
  const char *psa = slot + C::KP_SA_OFF + (wm * C::WM + l31) * 4;
  const char *psb = slot + C::SB_OFF + (wn * 64 + 4 * h) * 2;
  v4i af[TN][2];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    af[tn][0] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pw0 + tn * 2048));
    af[tn][1] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pw1 + tn * 2048));
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const v4i b0 = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pa0 + tm * 2048));
    const v4i b1 = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pa1 + tm * 2048));
    const float sa = (float)*reinterpret_cast<const half_t *>(psa + tm * 128);
    const float nms = -kMagic * sa;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      __builtin_amdgcn_sched_barrier(0);
      // each 64-column half is de-quantised on its own (the contract of include/atom_hip.h): t = round_f32(idot * sA8)
      v16i magic;
#pragma unroll
      for (int i = 0; i < 16; ++i) magic[i] = kMagicBits;
      v16i a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][0], b0, magic, 0, 0, 0);
      a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][1], b1, a, 0, 0, 0);
      v2u sbp[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) sbp[q] = *reinterpret_cast<const v2u *>(psb + (tn * 32 + 8 * q) * 2);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const half_t *hv = reinterpret_cast<const half_t *>(&sbp[r >> 2]);
        const float t = __builtin_fmaf(__int_as_float(a[r]), sa, nms);
        c[tn][tm][r] = __builtin_fmaf(t, (float)hv[r & 3], c[tn][tm][r]);
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

  // split-K (SK): blockIdx.y owns the K steps [s_begin, nsteps) and writes FP32 partial sums to p.ws
  const int total_steps = p.G + 2;
  const int s_begin = SK ? (int)((int64_t)total_steps * blockIdx.y / p.splits) : 0;
  const int nsteps = SK ? (int)((int64_t)total_steps * (blockIdx.y + 1) / p.splits) : total_steps;
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
  for (; step < nsteps; ++step)
    ATOM_F6_STEP(if (!(ABL & 1)) issue(step + NS - 1), (compute_keeper<C>(lds + (step % NS) * C::STAGE_BYTES, wm, wn, lane, c)))
#undef ATOM_F6_STEP
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

__device__ __forceinline__ void dequant4x(const v4f_t &acc, float sa, const v2u &sb, float (&c)[4]) {
  const half_t *hv = reinterpret_cast<const half_t *>(&sb);
  float t[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = acc[r] * sa;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    c[r] = __builtin_fmaf(t[r], (float)hv[r], c[r]);
    asm volatile("" : "+v"(c[r]));
  }
}

template <class C, class F = NoDma>
__device__ __forceinline__ void compute_int4_x16(const char *slot, int wm, int wn, int lane, float (&c)[4][C::WM / 16][4], F dma = F(),
                                                 bool older = false) {
  const int l15 = lane & 15, kb = lane >> 4;
  const char *pw = slot + (wn * 64 + l15) * PITCH + kb * 24;                  // + fb*16*PITCH
  constexpr int NTB = C::WM / 16;                                             // token blocks of the wave tile
  const char *pa = slot + C::A_OFF + (wm * C::WM + l15) * PITCH + kb * 24;    // + tb*16*PITCH
  const char *psa = slot + C::A_OFF + (wm * C::WM + l15) * PITCH + 96;        // + tb*16*PITCH
  const char *psb = slot + C::SB_OFF + (wn * 64 + 4 * kb) * 2;                // + fb*32
  v8i af[4], bf[2];
  v2u sb[4];
  // fragment loads in the order the MFMAs consume them (all 8 waves hit the LDS at once behind the barrier)
  bf[0] = frag24(pa);
  af[0] = frag24(pw);
  half_t sah = *reinterpret_cast<const half_t *>(psa);
#pragma unroll
  for (int fb = 1; fb < 4; ++fb) af[fb] = frag24(pw + fb * 16 * PITCH);
#pragma unroll
  for (int fb = 0; fb < 4; ++fb) sb[fb] = *reinterpret_cast<const v2u *>(psb + fb * 32);
  // tiles in pairs (fb 0,1 / 2,3 of a token block): a pair's MFMAs are issued one pair ahead of its 16 VALU instructions
  // (8 multiplies, then 8 FMAs: no dependent back-to-back issue)
  constexpr int GS = 2, NG = 4 * NTB / GS, GPB = 4 / GS, DEPTH = 1;
  v4f_t acc[DEPTH + 1][GS];
  auto mma = [&](int g) {
    const int tb = g / GPB, f0 = (g % GPB) * GS;
#pragma unroll
    for (int k = 0; k < GS; ++k) {
      acc[g % (DEPTH + 1)][k] = v4f_t{0.f, 0.f, 0.f, 0.f};
      acc[g % (DEPTH + 1)][k] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[f0 + k], bf[tb & 1], acc[g % (DEPTH + 1)][k], 3, 3, 0, 127, 0, 127);
    }
  };
  mma(0);
  float sa = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int tb = g / GPB, f0 = (g % GPB) * GS;
    __builtin_amdgcn_sched_barrier(0);
    if (g % GPB == 0) {
      sa = (float)sah;
      if (tb + 1 < NTB) {
        bf[(tb + 1) & 1] = frag24(pa + (tb + 1) * 16 * PITCH);
        sah = *reinterpret_cast<const half_t *>(psa + (tb + 1) * 16 * PITCH);
      }
      // the two waves of a SIMD swap priority mid-step (see compute_int4)
      if constexpr (C::NW >= 8) {
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
      float t[4 * GS];
#pragma unroll
      for (int k = 0; k < GS; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[4 * k + r] = acc[g % (DEPTH + 1)][k][r] * sa;
#pragma unroll
      for (int k = 0; k < GS; ++k) {
        const half_t *hv = reinterpret_cast<const half_t *>(&sb[f0 + k]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          c[f0 + k][tb][r] = __builtin_fmaf(t[4 * k + r], (float)hv[r], c[f0 + k][tb][r]);
          asm volatile("" : "+v"(c[f0 + k][tb][r]));
        }
      }
    }
  }
}

// keeper half-step on v_mfma_i32_16x16x64_i8, same tile layout; each half de-quantised on its own (the contract)
template <class C>
__device__ __forceinline__ void compute_keeper_x16(const char *slot, int wm, int wn, int lane, float (&c)[4][C::WM / 16][4]) {
  const int l15 = lane & 15, kb = lane >> 4;
  const int sw = (l15 >> 2) & 3;
  const char *pw = slot + (wn * 64 + l15) * 64 + ((kb ^ sw) << 4);                     // + fb*16*64
  constexpr int NTB = C::WM / 16;
  const char *pa = slot + (C::BN + wm * C::WM + l15) * 64 + ((kb ^ sw) << 4);          // + tb*16*64
  const char *psa = slot + C::KP_SA_OFF + (wm * C::WM + l15) * 4;                      // + tb*64
  const char *psb = slot + C::SB_OFF + (wn * 64 + 4 * kb) * 2;
  v4i af[4];
  v2u sb[4];
#pragma unroll
  for (int fb = 0; fb < 4; ++fb) {
    af[fb] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pw + fb * 1024));
    sb[fb] = *reinterpret_cast<const v2u *>(psb + fb * 32);
  }
#pragma unroll
  for (int tb = 0; tb < NTB; ++tb) {
    const v4i b = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pa + tb * 1024));
    const float sa = (float)*reinterpret_cast<const half_t *>(psa + tb * 64);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
      v4i a = {0, 0, 0, 0};
      a = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[fb], b, a, 0, 0, 0);
      const v4f_t f = {(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
      dequant4x(f, sa, sb[fb], c[fb][tb]);
    }
  }
}

template <class C, bool SK = false>
__global__ __launch_bounds__(C::NT, C::OCC) void gemm_w4a4_f6x16_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NS = C::NS;
  constexpr int NTB = C::WM / 16;
  static_assert(C::WM % 64 == 0 && NS >= 2, "x16: wave tiles of 64 features x 64 or 128 tokens");
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

  float c[4][NTB][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NTB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[a][b][r] = 0.f;

  // split-K (SK): blockIdx.y owns the K steps [s_begin, nsteps) and writes FP32 partial sums to p.ws
  const int total_steps = p.G + 2;
  const int s_begin = SK ? (int)((int64_t)total_steps * blockIdx.y / p.splits) : 0;
  const int nsteps = SK ? (int)((int64_t)total_steps * (blockIdx.y + 1) / p.splits) : total_steps;
  auto issue = [&](int step) {
    char *slot = lds + (step % NS) * C::STAGE_BYTES;
    const int s = min(step, nsteps - 1);
    if (s < p.G) issue_int4<C>(p, s, slot, wave, lane, m0, n0);
    else issue_keeper<C>(p, s - p.G, slot, wave, lane, m0, n0);
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s_begin + s);
  const bool older = wave < C::NW / 2;
  int step = s_begin;
  for (; step + NS - 1 < min(p.G, nsteps); ++step) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::GLDS * (NS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    char *nslot = lds + ((step + NS - 1) % NS) * C::STAGE_BYTES;
    const int g = step + NS - 1;
    auto dma = [&](int i) { issue_int4_piece<C>(p, g, nslot, wave, lane, m0, n0, i); };
    __builtin_amdgcn_sched_barrier(0);
    compute_int4_x16<C>(lds + (step % NS) * C::STAGE_BYTES, wm, wn, lane, c, dma, older);
  }
  for (; step < min(p.G, nsteps); ++step) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::GLDS * (NS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(step + NS - 1);
    __builtin_amdgcn_sched_barrier(0);
    compute_int4_x16<C>(lds + (step % NS) * C::STAGE_BYTES, wm, wn, lane, c, NoDma(), older);
  }
  __builtin_amdgcn_s_setprio(0);
  for (; step < nsteps; ++step) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::GLDS * (NS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(step + NS - 1);
    __builtin_amdgcn_sched_barrier(0);
    compute_keeper_x16<C>(lds + (step % NS) * C::STAGE_BYTES, wm, wn, lane, c);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int l15 = lane & 15, kb = lane >> 4;
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
  // epilogue: per wave [64 tokens][64 features] fp16 through LDS (row stride 144 B), halves of 64 tokens
  constexpr int EP_STRIDE = 144;
  char *ep = lds + wave * (64 * EP_STRIDE);
#pragma unroll
  for (int half = 0; half < C::WM / 64; ++half) {
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const int tb = half * 4 + t4;
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
template <class C, bool LAST, class FS, class FD>
__device__ __forceinline__ void p_step(PRegs<C> &R, const char *slot, const char *nslot, int wm, int wn, int lane,
                                       float (&c)[4][8][4], FS sync, FD mid) {
  const int l15 = lane & 15, kb = lane >> 4;
  const char *pw = slot + wn * 64 * PITCH, *pa = slot + C::A_OFF + wm * 128 * PITCH;
  const char *npw = nslot + wn * 64 * PITCH, *npa = nslot + C::A_OFF + wm * 128 * PITCH;
  const char *psa = pa + l15 * (2 * PITCH) + 96, *npsa = npa + l15 * (2 * PITCH) + 96;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int tb = p_tb(i), h = p_h(i);
    __builtin_amdgcn_sched_barrier(0);
    if (i == 8) sync();
    if (h == 0) R.sa[tb] = (float)R.sah[p_buf(tb)];         // this block's token scale arrived with its fragment
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 2; ++k)
      R.acc[i & 1][k] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(R.af[2 * h + k], R.bf[p_buf(tb)], v4f_t{0.f, 0.f, 0.f, 0.f}, 3, 3, 0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- loads behind this slot's MFMAs, into registers whose last reader has just issued
    if (h == 1 && i < 12 && tb + 2 < 8) {                    // token block tb + 2 (the buffer of block tb; block 0: of block 6
      R.bf[p_buf(tb + 2)] = frag3<C>(R, pa, p_row(tb + 2) * PITCH);   // of the previous step)
      R.sah[p_buf(tb + 2)] = *reinterpret_cast<const half_t *>(psa + p_row(tb + 2) * PITCH);
    }
    if constexpr (!LAST) {
      if (i == 10) p_load_sb<C>(R, nslot, wn, kb);
      if (i == 12) {                                         // next step's block 0 (buffer 2: free since slot 1)
        R.bf[2] = frag3<C>(R, npa, p_row(0) * PITCH);
        R.sah[2] = *reinterpret_cast<const half_t *>(npsa + p_row(0) * PITCH);
      }
      if (i == 13) {                                         // fragments fb 0,1: last read by this slot
        R.af[0] = frag3<C>(R, npw, p_row(0) * PITCH);
        R.af[1] = frag3<C>(R, npw, p_row(1) * PITCH);
      }
      if (i == 15) {
        R.af[2] = frag3<C>(R, npw, p_row(2) * PITCH);
        R.af[3] = frag3<C>(R, npw, p_row(3) * PITCH);
        R.bf[1] = frag3<C>(R, npa, p_row(1) * PITCH);        // next step's block 1
        R.sah[1] = *reinterpret_cast<const half_t *>(npsa + p_row(1) * PITCH);
      }
    }
    if (i >= 8) mid(i - 8);
    __builtin_amdgcn_sched_barrier(0);
    // ---- de-quantisation of the previous slot's pair (slot 15 of the previous step for i == 0)
    {
      const int j = (i + 15) & 15, dtb = p_tb(j), dh = p_h(j);
      const float sa = R.sa[dtb];
      float t[8];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[4 * k + r] = R.acc[j & 1][k][r] * sa;
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          c[2 * dh + k][dtb][r] = __builtin_fmaf(t[4 * k + r], R.sb[2 * dh + k][r], c[2 * dh + k][dtb][r]);
          asm volatile("" : "+v"(c[2 * dh + k][dtb][r]));
        }
    }
    // the next stage's weight scales replace a pair's FP32 copies once its last de-quantisation of this step is done:
    // pair 0 (slot 13 = block 7) after slot 14's de-quantisation, pair 1 after the carried one in the next step's slot 0
    if constexpr (!LAST) { if (i == 14) p_cvt_sb<C>(R, 0); }
    if (i == 0) p_cvt_sb<C>(R, 1);
  }
}

// the carried pair (slot 15) of the last int4 step
template <class C>
__device__ __forceinline__ void p_drain(PRegs<C> &R, float (&c)[4][8][4]) {
  const float sa = R.sa[7];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[2 + k][7][r] = __builtin_fmaf(R.acc[1][k][r] * sa, R.sb[2 + k][r], c[2 + k][7][r]);
}

// keeper half-step on v_mfma_i32_16x16x64_i8 with the interleaved row mapping; each half de-quantised on its own (the contract)
template <class C>
__device__ __forceinline__ void p_keeper(const char *slot, int wm, int wn, int lane, float (&c)[4][8][4]) {
  const int l15 = lane & 15, kb = lane >> 4;
  const int sw = (l15 >> 1) & 3;                                       // swizzle key (row >> 2) & 3 of row 2 * l15 + (blk & 1)
  const char *pw = slot + (wn * 64 + 2 * l15) * 64 + ((kb ^ sw) << 4);
  const char *pa = slot + (C::BN + wm * 128 + 2 * l15) * 64 + ((kb ^ sw) << 4);
  const char *psa = slot + C::KP_SA_OFF + (wm * 128 + 2 * l15) * 4;
  const char *psb = slot + C::SB_OFF + (wn * 64 + 8 * kb) * 2;
  v4i af[4];
  v4u sbp[2];
#pragma unroll
  for (int fb = 0; fb < 4; ++fb) af[fb] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pw + p_row(fb) * 64));
  sbp[0] = *reinterpret_cast<const v4u *>(psb);
  sbp[1] = *reinterpret_cast<const v4u *>(psb + 64);
#pragma unroll
  for (int tb = 0; tb < 8; ++tb) {
    const v4i b = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pa + p_row(tb) * 64));
    const float sa = (float)*reinterpret_cast<const half_t *>(psa + p_row(tb) * 4);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
      __builtin_amdgcn_sched_barrier(0);
      v4i a = {0, 0, 0, 0};
      a = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[fb], b, a, 0, 0, 0);
      const half_t *hv = reinterpret_cast<const half_t *>(&sbp[fb >> 1]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = (float)a[r] * sa;
        c[fb][tb][r] = __builtin_fmaf(t, (float)hv[2 * r + (fb & 1)], c[fb][tb][r]);
        asm volatile("" : "+v"(c[fb][tb][r]));
      }
    }
  }
}

template <class C>
__global__ __launch_bounds__(C::NT, C::OCC) void gemm_w4a4_f6p_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
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
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::GLDS) : "memory");
  __builtin_amdgcn_s_barrier();
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
    __builtin_amdgcn_s_barrier();
  };
  // Steps 0 .. G-1.  The LDS-DMA of stage s + 2 goes out behind the mid-step barrier of step s: int4 stages one instruction
  // per pair slot, the keeper halves (the last two stages) in one block.
  {
    int s = 0;
    auto dma4 = [&](int i) { if (i < C::GLDS) issue_int4_piece<C>(p, s + 2, slot_of(s + 2), wave, lane, m0, n0, i); };
    auto dmak0 = [&](int i) { if (i == 0) issue_keeper<C>(p, 0, slot_of(G), wave, lane, m0, n0); };
    auto dmak1 = [&](int i) { if (i == 0) issue_keeper<C>(p, 1, slot_of(G + 1), wave, lane, m0, n0); };
    for (; s + 2 < G; ++s) p_step<C, false>(R, slot_of(s), slot_of(s + 1), wm, wn, lane, c, sync, dma4);
    if (G >= 2) { p_step<C, false>(R, slot_of(s), slot_of(s + 1), wm, wn, lane, c, sync, dmak0); ++s; }
    p_step<C, true>(R, slot_of(s), slot_of(s + 1), wm, wn, lane, c, sync, dmak1);
  }
  p_drain<C>(R, c);
  // keeper half 0 was published by the last mid-step barrier; half 1 was issued behind it
  p_keeper<C>(slot_of(G), wm, wn, lane, c);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  p_keeper<C>(slot_of(G + 1), wm, wn, lane, c);

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
      *reinterpret_cast<v4u *>(p.D + (int64_t)m * p.N + n) = o;
    }
  }
}

template <class C>
static int launch_p(const GemmParams &p, hipStream_t s) {
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_w4a4_f6p_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          C::LDS_BYTES) != hipSuccess)
    return ATOM_ERR_LAUNCH;
  const int nbm = (p.M + C::BM - 1) / C::BM, nbn = (p.N + C::BN - 1) / C::BN;
  hipLaunchKernelGGL((gemm_w4a4_f6p_kernel<C>), dim3((unsigned)(nbm * nbn)), dim3(C::NT), C::LDS_BYTES, s, p);
  return check_launch();
}

template <class C, bool SK = false>
static int launch_x16(const GemmParams &p, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_w4a4_f6x16_kernel<C, SK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
      return ATOM_ERR_LAUNCH;
    attr_set = true;
  }
  const int nbm = (p.M + C::BM - 1) / C::BM, nbn = (p.N + C::BN - 1) / C::BN;
  hipLaunchKernelGGL((gemm_w4a4_f6x16_kernel<C, SK>), dim3((unsigned)(nbm * nbn), (unsigned)(SK ? p.splits : 1)), dim3(C::NT),
                     C::LDS_BYTES, s, p);
  if (SK) {
    const int64_t MN = (int64_t)p.M * p.N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN / 8 + 255) / 256)), dim3(256), 0, s, p.ws, p.D, MN, p.splits);
  }
  return check_launch();
}

template <class C, bool SK, int ABL = 0>
static int launch(const GemmParams &p, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_w4a4_f6_kernel<C, SK, ABL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
      return ATOM_ERR_LAUNCH;
    attr_set = true;
  }
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

// cfg: 0 = 256x256 (8 waves) and 3 = 128x128 (4 waves, three workgroups per CU) on 16x16x128 MFMA micro-tiles, 2 = 64x128
// (2 waves, 32x32x64 MFMA; split-K when p.splits > 1 and p.ws is set); tuning only: 1 = 256x128, 10 / 13 = 256x256 /
// 128x128 on the 32x32x64 MFMA
int launch_gemm_f6(const GemmParams &p, int cfg, hipStream_t s) {
#ifdef ATOM_F6_ABLATE
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
#endif
  if (cfg == 2) {
    if (p.splits > 1 && p.ws) return f6::launch<f6::Cfg<64, 128, 2, 3>, true>(p, s);
    return f6::launch<f6::Cfg<64, 128, 2, 3>, false>(p, s);
  }
  if (cfg == 10) return f6::launch<f6::Cfg<256, 256, 4, 3>, false>(p, s);   // tuning: 256x256 on the 32x32x64 MFMA
  if (cfg == 1) return f6::launch<f6::Cfg<256, 128, 4, 2>, false>(p, s);
  if (cfg == 13) return f6::launch<f6::Cfg<128, 128, 2, 2, 3>, false>(p, s);  // tuning: 128x128 on the 32x32x64 MFMA
  if (cfg == 3) {                                                              // 4 waves, three workgroups per CU
    if (p.splits > 1 && p.ws) return f6::launch_x16<f6::Cfg<128, 128, 2, 2, 3>, true>(p, s);
    return f6::launch_x16<f6::Cfg<128, 128, 2, 2, 3>>(p, s);
  }
  if (cfg == 30) return f6::launch_p<f6::Cfg<256, 256, 4, 3>>(p, s);          // 256x256, pipelined across K steps
  return f6::launch_x16<f6::Cfg<256, 256, 4, 3>>(p, s);        // 256x256 on the 16x16x128 MFMA
}

}  // namespace atom
