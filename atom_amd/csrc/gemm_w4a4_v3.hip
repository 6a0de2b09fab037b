// W4A4 GEMM, prefill kernel v3 = gemm_w4a4_v2.hip with the block geometry as a template parameter (see that file's
// header for the design).  The second instantiation runs TWO independent 4-wave workgroups per CU (256x128 tile, 76 KiB
// LDS each) instead of one 8-wave workgroup: the two workgroups drift out of phase, so one's prologue / barrier /
// epilogue stalls are covered by the other's main loop.
#include "common.h"

namespace atom {
namespace v3 {

constexpr float kMagic = 12582912.0f;
constexpr int kMagicBits = 0x4B400000;
constexpr int TN = 2;                          // wave tile (32*TM)(m) x 64(n); TM is a Cfg parameter (4, or 2 for skinny M)

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// AW ("wide activations"): the activation operand arrives already widened to int8 (code * 16), 128 bytes per token and
// group, with the even / odd channels of every 32-channel block de-interleaved into two 16-byte chunks -- exactly the
// two operands widen() makes from one packed chunk -- so the kernel skips 3.0 of its 4.5 widening VALU per MFMA.
// Produced by the activation-quant kernels (ATOM_QUANT_WIDE_CODES); the weights stay packed INT4.
template <int BM_, int BN_, int NS_, int TM_ = 4, bool AW_ = false, bool PAIR_ = false>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, NS = NS_, TM = TM_;
  static constexpr bool AW = AW_;
  // ATOM_B_SCALE_PAIRS (the caller asserts weight_channel_group = 2): the int4 steps form one scale product per channel pair -- 2.5
  // instead of 3 de-quantisation VALU per element; only the geometries of large batches are instantiated with it
  static constexpr bool PAIR = PAIR_;
  static constexpr int WM = 32 * TM;                         // rows of activations per wave
  static constexpr int WGM = BM / WM, WGN = BN / 64, NW = WGM * WGN, NT = NW * 64;
  static constexpr int A_ROWB = AW ? 128 : 64;               // bytes per activation row and stage
  static constexpr int GW = BN / 16;                         // 1 KiB DMA granules: 16 weight rows ...
  static constexpr int GA = BM * A_ROWB / 1024;              // ... 16 packed / 8 wide activation rows
  static constexpr int A_OFF = BN * 64;                      // weights rows first, then the activation rows
  static constexpr int DATA_BYTES = (GW + GA) * 1024;
  static constexpr int SB_BYTES = (BN < 128 ? 128 : BN) * 2; // BN fp16, dense (one dword DMA always moves 128 of them)
  static constexpr int SB_OFF = DATA_BYTES;
  static constexpr int SA_OFF = DATA_BYTES + SB_BYTES;       // BM dwords (fp16 in the low half)
  static constexpr int STAGE_BYTES = DATA_BYTES + SB_BYTES + BM * 4;
  static constexpr int IPW = (GW + GA) / NW;                 // data DMA instructions per wave per stage
  static constexpr int NSA = BM / 64, NSB = (BN + 127) / 128; // scale DMA instructions per stage (ushort / dword)
  static constexpr int SPW = (NSA + NSB + NW - 1) / NW;      // scale DMA slots per wave (padded with duplicates)
  static constexpr int GLDS = IPW + SPW;
  static constexpr int EP_BYTES = NW * 64 * 144;
  static constexpr int LDS_BYTES = NS * STAGE_BYTES > EP_BYTES ? NS * STAGE_BYTES : EP_BYTES;
  static_assert((GW + GA) % NW == 0 && BN % 64 == 0 && BM % 64 == 0 && BM % WM == 0 && (TM == 2 || TM == 4), "geometry");
  static_assert(STAGE_BYTES % 16 == 0, "stage alignment");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <class C>
struct StageAddr {
  unsigned idx[C::IPW];       // clamped global row (feature n or token m) this lane fetches for granule i
  unsigned jw, ja, ja1;       // 16 * logical chunk this lane's LDS slot must receive (weights / activation granules;
                              // wide activation granules: even / odd granule)
  unsigned scale[C::SPW];
};

template <class C>
__device__ __forceinline__ void make_stage_addr(const GemmParams &p, int wave, int lane, int m0, int n0, StageAddr<C> &a) {
  // XOR swizzle: granule rows are 16-aligned (8-aligned for wide rows), so the logical chunk only depends on the lane
  a.jw = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  // wide rows are 128 B = half of the 64 LDS banks: the 16 lanes of a ds_read_b128 phase read rows r..r+15, two rows
  // per 256-byte bank line, so the key is (row >> 1) & 7 (with row & 7, rows r and r+8 collide: 12x the conflicts)
  a.ja = C::AW ? (unsigned)(((lane & 7) ^ ((lane >> 4) & 3)) * 16) : a.jw;
  a.ja1 = C::AW ? (unsigned)(((lane & 7) ^ (4 | ((lane >> 4) & 3))) * 16) : a.jw;
#pragma unroll
  for (int i = 0; i < C::IPW; ++i) {
    const int gidx = wave * C::IPW + i;                       // 1 KiB granule inside the stage (wave-uniform)
    const bool isW = gidx < C::GW;
    if (isW || !C::AW) {
      const int row = gidx * 16 + (lane >> 2);                // 16 rows x 64 B; weights rows [0,BN), then activations
      a.idx[i] = (unsigned)(isW ? min(n0 + row, p.N - 1) : min(m0 + row - C::BN, p.M - 1));
    } else {
      const int ra = (gidx - C::GW) * 8 + (lane >> 3);        // 8 rows x 128 B of wide activations
      a.idx[i] = (unsigned)min(m0 + ra, p.M - 1);
    }
  }
#pragma unroll
  for (int s = 0; s < C::SPW; ++s) {
    const int slot = wave * C::SPW + s;
    if (slot < C::NSA) {
      const int idx = min(m0 + slot * 64 + lane, p.M - 1);
      a.scale[s] = p.ref_layout ? ref_scale_index(idx) : idx;
    } else {
      const int part = slot - C::NSA < C::NSB ? slot - C::NSA : 0;
      a.scale[s] = min(n0 + part * 128 + 2 * lane, p.N - 2);
    }
  }
}

template <class C>
__device__ __forceinline__ void issue_stage(const GemmParams &p, int step, char *slot, int wave, const StageAddr<C> &a) {
  const bool int4 = step < p.G;
  const int koff = (int4 ? step : step - p.G) * 64;
  const uint8_t *wb = (int4 ? p.B4 : p.B8) + koff;
  const uint8_t *ab = (int4 ? p.A4 + (C::AW ? koff : 0) : p.A8) + koff;   // wide rows advance 128 B per group
  // byte offset of a lane's 16-byte chunk = row * row stride + 16 * logical chunk; in the keeper half-steps a wide
  // activation row still fills 8 slots, logical chunks 4..7 with copies of 0..3 (never read)
  const unsigned strideW = int4 ? (unsigned)p.K4h : (unsigned)kKeeper;
  const unsigned strideA = int4 ? (unsigned)(C::AW ? 2 * p.K4h : p.K4h) : (unsigned)kKeeper;
  const unsigned kmask = (C::AW && !int4) ? 0x30u : 0x70u;
  const unsigned sl0 = lds_addr(slot);                     // one generic -> LDS cast per stage, integer offsets per piece (common.h)
#pragma unroll
  for (int i = 0; i < C::IPW; ++i) {
    const int gidx = wave * C::IPW + i;
    const bool isW = gidx < C::GW;
    const uint8_t *base = isW ? wb : ab;
    const unsigned ja = (((gidx - C::GW) & 1) ? a.ja1 : a.ja) & kmask;
    const unsigned off = a.idx[i] * (isW ? strideW : strideA) + (isW ? a.jw : ja);
    lds_dma_sv<16>(base, off, sl0 + gidx * 1024);
  }
  const bool keeper = step >= p.G;
  const int g = min(step, p.G - 1);
  const half_t *sAb = keeper ? p.sA8 : p.sA + (int64_t)g * p.ldA;
  const half_t *sBb = keeper ? p.sB8 : p.sB + (int64_t)g * p.N;
#pragma unroll
  for (int s = 0; s < C::SPW; ++s) {
    const int sl = wave * C::SPW + s;
    if (sl < C::NSA) {   // one fp16 per lane; lands as one zero-extended dword per lane (tools/probes/glds_probe.cpp)
      lds_dma_sv<2>(sAb, (unsigned)a.scale[s] * 2u, sl0 + C::SA_OFF + sl * 256);
    } else {             // a dword = two adjacent channels per lane, dense fp16 image (padding slots repeat part 0)
      const int part = sl - C::NSA < C::NSB ? sl - C::NSA : 0;
      lds_dma_sv<4>(sBb, (unsigned)a.scale[s] * 2u, sl0 + C::SB_OFF + part * 256);
    }
  }
}

__device__ __forceinline__ void widen(const v4u p, v4i &lo, v4i &hi) {
  const v4u l = (p << 4) & 0xF0F0F0F0u;
  const v4u h = p & 0xF0F0F0F0u;
  lo = __builtin_bit_cast(v4i, l);
  hi = __builtin_bit_cast(v4i, h);
}

// Loop-invariant per-lane byte offsets inside a stage.  Every LDS address of the K loop is then
//   stage base (one v_add per step)  +  one of these VGPRs  +  a compile-time immediate (tile * 2048, region offsets),
// instead of ~20 v_add_u32 per step: with a VALU instruction costing the SIMD 4 cycles that is 4 % of the kernel.
struct LaneOff {
  int w0, w1;     // weight rows:      (wn*64 + l31)*64 + swizzled chunk (0+h) / (2+h)
  int a0, a1;     // activation rows:  (BN + wm*WM + l31)*64 + swizzled chunk
  int a2, a3;     // wide activations: A_OFF + (wm*WM + l31)*128 + swizzled chunks 2h, 2h+1 (a0,a1) and 4+2h, 5+2h (a2,a3)
  int ak0, ak1;   // wide activations, keeper half-steps: logical chunks 0+h, 2+h
  int sa;         // activation scale: SA_OFF + (wm*WM + l31)*4
  int sb;         // weight scales:    SB_OFF + (wn*64 + 4h)*2
};

template <class C>
__device__ __forceinline__ LaneOff make_lane_off(int wm, int wn, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
  const int sw = (l31 >> 2) & 3;                 // tile offsets are multiples of 32 rows: they do not touch bits 2,3
  LaneOff o;
  o.w0 = (wn * 64 + l31) * 64 + (((0 + h) ^ sw) << 4);
  o.w1 = (wn * 64 + l31) * 64 + (((2 + h) ^ sw) << 4);
  if constexpr (C::AW) {
    const int base = C::A_OFF + (wm * C::WM + l31) * 128, s8 = (l31 >> 1) & 7;
    o.a0 = base + (((2 * h) ^ s8) << 4);
    o.a1 = base + (((2 * h + 1) ^ s8) << 4);
    o.a2 = base + (((4 + 2 * h) ^ s8) << 4);
    o.a3 = base + (((5 + 2 * h) ^ s8) << 4);
    o.ak0 = base + (((0 + h) ^ s8) << 4);
    o.ak1 = base + (((2 + h) ^ s8) << 4);
  } else {
    o.a0 = (C::BN + wm * C::WM + l31) * 64 + (((0 + h) ^ sw) << 4);
    o.a1 = (C::BN + wm * C::WM + l31) * 64 + (((2 + h) ^ sw) << 4);
    o.a2 = o.a3 = o.ak0 = o.ak1 = 0;
  }
  o.sa = C::SA_OFF + (wm * C::WM + l31) * 4;
  o.sb = C::SB_OFF + (wn * 64 + 4 * h) * 2;
  return o;
}

template <bool INT4>
__device__ __forceinline__ void load_frag(const char *p0, const char *p1, v4i (&f)[4]) {
  const v4u c0 = *reinterpret_cast<const v4u *>(p0);
  const v4u c1 = *reinterpret_cast<const v4u *>(p1);
  if constexpr (INT4) {
    widen(c0, f[0], f[1]);
    widen(c1, f[2], f[3]);
  } else {
    f[0] = __builtin_bit_cast(v4i, c0);
    f[1] = __builtin_bit_cast(v4i, c1);
  }
}

// De-quantise one 32x32 tile: acc holds the magic-biased FP32 image of the integer dot products -- the register read as a float is
// 12582912 + idot exactly, so idot = acc - 12582912 is one exact v_sub (no quarter-rate v_cvt_f32_i32).  The contract (round 5,
// include/atom_hip.h): s = sA * sB -- exact in FP32 --, c = fma(idot, s, c).  (Rounds 1-4: t = fma(acc, sA, -12582912 sA), c = fma(t, sB, c)
// -- 2 instead of 3 instructions per element here, but two roundings and no shared scale product for the BF6 kernels the library
// ships as its fast path; this INT8 kernel serves the plain entry point only.)  Four elements at a time: the wave tile leaves no
// registers for 16 products.
template <bool PAIR = false>
__device__ __forceinline__ void dequant16_magic(const v16i &a, float sa, const v2u (&sbp)[4], float (&c)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const half_t *hv = reinterpret_cast<const half_t *>(&sbp[q]);
    float s[4], t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = (PAIR && (r & 1)) ? s[r - 1] : (float)hv[r] * sa;   // (a lane's four are consecutive channels)
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = __int_as_float(a[4 * q + r]) - kMagic;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      c[4 * q + r] = __builtin_fmaf(t[r], s[r], c[4 * q + r]);
      asm volatile("" : "+v"(c[4 * q + r]));
    }
  }
}

template <class C, bool INT4>
__device__ __forceinline__ void compute_step(const char *slot, const LaneOff &lo, float (&c)[TN][C::TM][16]) {
  constexpr int TM = C::TM;
  constexpr int KS = INT4 ? 4 : 2;
  constexpr bool AW = C::AW;
  constexpr int ASTR = AW ? 4096 : 2048;                 // bytes between consecutive 32-token tiles
  const char *pw0 = slot + lo.w0, *pw1 = slot + lo.w1;
  const char *pa0 = slot + (AW && !INT4 ? lo.ak0 : lo.a0), *pa1 = slot + (AW && !INT4 ? lo.ak1 : lo.a1);
  const char *pa2 = slot + lo.a2, *pa3 = slot + lo.a3;
  const char *psa = slot + lo.sa, *psb = slot + lo.sb;
  v16i magic;
#pragma unroll
  for (int i = 0; i < 16; ++i) magic[i] = kMagicBits;
  v4i af[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) load_frag<INT4>(pw0 + tn * 2048, pw1 + tn * 2048, af[tn]);

  if constexpr (AW) {
    // Wide activations need no VALU between LDS and the MFMA, so the 4 (keeper: 2) K slices of a token tile live in
    // ONE set of registers that is refilled in place: as soon as the last chain of tile tm has consumed slice s, the
    // slice s of tile tm+1 is requested into the same registers and lands under that chain's de-quantisation.
    const char *pa[4] = {pa0, pa1, pa2, pa3};
    v4i bf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) bf[s] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pa[s]));
    half_t sah = *reinterpret_cast<const half_t *>(psa);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      __builtin_amdgcn_sched_barrier(0);
      const float sa = (float)sah * (INT4 ? (1.0f / 256.0f) : 1.0f);
      if (tm + 1 < TM) sah = *reinterpret_cast<const half_t *>(psa + (tm + 1) * 128);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        __builtin_amdgcn_sched_barrier(0);
        v16i a = magic;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][s], bf[s], a, 0, 0, 0);
          if (tn == TN - 1 && tm + 1 < TM) {
            __builtin_amdgcn_sched_barrier(0);             // keep the refill BEHIND the MFMA that reads the old value
            bf[s] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(pa[s] + (tm + 1) * ASTR));
          }
        }
        v2u sbp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          sbp[q] = *reinterpret_cast<const v2u *>(psb + (tn * 32 + 8 * q) * 2);
        dequant16_magic<C::PAIR && INT4>(a, sa, sbp, c[tn][tm]);
      }
    }
    return;
  }

  v4u pk0, pk1;
  half_t sah;
  auto request = [&](int tm) {
    pk0 = *reinterpret_cast<const v4u *>(pa0 + tm * ASTR);
    pk1 = *reinterpret_cast<const v4u *>(pa1 + tm * ASTR);
    sah = *reinterpret_cast<const half_t *>(psa + tm * 128);
  };
  request(0);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    __builtin_amdgcn_sched_barrier(0);
    v4i bf[4];
    if constexpr (INT4) {
      widen(pk0, bf[0], bf[1]);
      widen(pk1, bf[2], bf[3]);
    } else {
      bf[0] = __builtin_bit_cast(v4i, pk0); bf[1] = __builtin_bit_cast(v4i, pk1); bf[2] = bf[0]; bf[3] = bf[1];
    }
    const float sa = (float)sah * (INT4 ? (1.0f / 256.0f) : 1.0f);
    if (tm + 1 < TM) request(tm + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      __builtin_amdgcn_sched_barrier(0);
      v16i a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][0], bf[0], magic, 0, 0, 0);
#pragma unroll
      for (int s = 1; s < KS; ++s) a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][s], bf[s], a, 0, 0, 0);
      v2u sbp[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        sbp[q] = *reinterpret_cast<const v2u *>(psb + (tn * 32 + 8 * q) * 2);
      dequant16_magic<C::PAIR && INT4>(a, sa, sbp, c[tn][tm]);
    }
  }
}


// The keeper: its 128 INT8 columns arrive as TWO stages of 64-byte rows (the int4 stage layout; slot0 = columns 0..63, slot1 =
// 64..127) and are multiplied in ONE step -- four chained MFMAs per 32x32 tile into one accumulator, one de-quantisation -- as the
// reference kernel does (Dense_layer_gemm_i4_o16.cuh:640-691; rounds 1-2 de-quantised the two halves separately).
template <class C>
__device__ __forceinline__ void compute_keeper(const char *slot0, const char *slot1, const LaneOff &lo, float (&c)[TN][C::TM][16]) {
  constexpr int TM = C::TM;
  constexpr bool AW = C::AW;
  constexpr int ASTR = AW ? 4096 : 2048;
  const char *psa = slot0 + lo.sa, *psb = slot0 + lo.sb;   // (both stages carry the same scales)
  const int a0 = AW ? lo.ak0 : lo.a0, a1 = AW ? lo.ak1 : lo.a1;
  v16i magic;
#pragma unroll
  for (int i = 0; i < 16; ++i) magic[i] = kMagicBits;
  v4i af[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    af[tn][0] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot0 + lo.w0 + tn * 2048));
    af[tn][1] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot0 + lo.w1 + tn * 2048));
    af[tn][2] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot1 + lo.w0 + tn * 2048));
    af[tn][3] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot1 + lo.w1 + tn * 2048));
  }
  v4i bf[4];
  half_t sah;
  auto request = [&](int tm) {
    bf[0] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot0 + a0 + tm * ASTR));
    bf[1] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot0 + a1 + tm * ASTR));
    bf[2] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot1 + a0 + tm * ASTR));
    bf[3] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(slot1 + a1 + tm * ASTR));
    sah = *reinterpret_cast<const half_t *>(psa + tm * 128);
  };
  request(0);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    __builtin_amdgcn_sched_barrier(0);
    const v4i b0 = bf[0], b1 = bf[1], b2 = bf[2], b3 = bf[3];
    const float sa = (float)sah;
    if (tm + 1 < TM) request(tm + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      __builtin_amdgcn_sched_barrier(0);
      v16i a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][0], b0, magic, 0, 0, 0);
      a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][1], b1, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][2], b2, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][3], b3, a, 0, 0, 0);
      v2u sbp[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) sbp[q] = *reinterpret_cast<const v2u *>(psb + (tn * 32 + 8 * q) * 2);
      dequant16_magic(a, sa, sbp, c[tn][tm]);
    }
  }
}

template <class C, bool TRACE = false, bool SK = false>
__global__ __launch_bounds__(C::NT, 2) void gemm_w4a4_v3_kernel(GemmParams p) {   // <= 256 VGPRs: two waves per SIMD
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  constexpr int TM = C::TM;
  float c[TN][TM][16];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[a][b][r] = 0.f;

  // K steps: the G int4 groups, then the keeper = G + 1 compute steps; DMA stages: G + 2 (the keeper's two 64-byte halves).
  // split-K (SK): blockIdx.y owns the compute steps [cb, ce) and writes FP32 partial sums to p.ws
  const int csteps = p.G + 1;
  const int cb = SK ? (int)((int64_t)csteps * blockIdx.y / p.splits) : 0;
  const int ce = SK ? (int)((int64_t)csteps * (blockIdx.y + 1) / p.splits) : csteps;
  const bool has_keeper = ce == csteps;
  const int s_begin = cb, nsteps = ce + (has_keeper ? 1 : 0);   // DMA stages [s_begin, nsteps)
  StageAddr<C> sa_;
  make_stage_addr<C>(p, wave, lane, m0, n0, sa_);
  const LaneOff lo_ = make_lane_off<C>(wm, wn, lane);
#pragma unroll
  for (int s = 0; s < C::NS - 1; ++s)
    issue_stage<C>(p, min(s_begin + s, nsteps - 1), lds + ((s_begin + s) % C::NS) * C::STAGE_BYTES, wave, sa_);

  // TRACE (tuning only): s_memtime stamps of block 0 into p.Dsz as u64[wave][step][4]
  unsigned long long *trace = reinterpret_cast<unsigned long long *>(p.Dsz);
  const bool tr = TRACE && blockIdx.x == 0 && lane == 0;
#define ATOM_V3_STEP(INT4)                                                                                           \
  {                                                                                                                  \
    if (tr) trace[(wave * 64 + step) * 4 + 0] = __builtin_amdgcn_s_memtime();                                        \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::GLDS * (C::NS - 2)) : "memory");                                     \
    if (tr) trace[(wave * 64 + step) * 4 + 1] = __builtin_amdgcn_s_memtime();                                        \
    __builtin_amdgcn_s_barrier();                                                                                    \
    if (tr) trace[(wave * 64 + step) * 4 + 2] = __builtin_amdgcn_s_memtime();                                        \
    issue_stage<C>(p, min(step + C::NS - 1, nsteps - 1), lds + ((step + C::NS - 1) % C::NS) * C::STAGE_BYTES, wave,  \
                   sa_);                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    if (tr) trace[(wave * 64 + step) * 4 + 3] = __builtin_amdgcn_s_memtime();                                        \
    compute_step<C, INT4>(lds + (step % C::NS) * C::STAGE_BYTES, lo_, c);                                            \
  }
  int step = s_begin;
  for (; step < min(p.G, nsteps); ++step) ATOM_V3_STEP(true)
#undef ATOM_V3_STEP
  if (has_keeper) {                                        // step == G: both halves (stages G, G + 1) in ONE compute step
    if (tr) trace[(wave * 64 + step) * 4 + 0] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (C::NS == 2) {                            // two-stage ring: the second half only fits now that stage G - 1 is read
      issue_stage<C>(p, p.G + 1, lds + ((p.G + 1) % C::NS) * C::STAGE_BYTES, wave, sa_);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    compute_keeper<C>(lds + (p.G % C::NS) * C::STAGE_BYTES, lds + ((p.G + 1) % C::NS) * C::STAGE_BYTES, lo_, c);
    ++step;
  }
  if (tr) trace[(wave * 64 + step) * 4 + 0] = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  if constexpr (SK) {
    // FP32 partial tile: lane owns token m and 4 consecutive features per (tile, q) -> one 16-byte store each
    const int l31 = lane & 31, h = lane >> 5;
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
          v4f o = {c[tn][tm][4 * q], c[tn][tm][4 * q + 1], c[tn][tm][4 * q + 2], c[tn][tm][4 * q + 3]};
          *reinterpret_cast<v4f *>(wsp + (int64_t)m * p.N + n) = o;
        }
    }
    return;
  }
  constexpr int EP_STRIDE = 144;
  char *ep = lds + wave * (64 * EP_STRIDE);
  const int l31 = lane & 31, h = lane >> 5;
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

}  // namespace v3

// D[m][n] = half( sum over splits, in split order, of the FP32 partials ): 8 features per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *ws, half_t *D, int64_t MN, int splits) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= MN) return;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int s = 0; s < splits; ++s) {
    const v4f a = *reinterpret_cast<const v4f *>(ws + s * MN + i);
    const v4f b = *reinterpret_cast<const v4f *>(ws + s * MN + i + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { acc[k] += a[k]; acc[4 + k] += b[k]; }
  }
  v4u o;
  half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
  for (int k = 0; k < 8; ++k) ov[k] = f2h(acc[k]);
  *reinterpret_cast<v4u *>(D + i) = o;
}

template <class C>
static int launch_v3_splitk(const GemmParams &p, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  if (ensure_max_lds(reinterpret_cast<const void *>(&v3::gemm_w4a4_v3_kernel<C, false, true>), C::LDS_BYTES, attr_done) != ATOM_OK) return ATOM_ERR_LAUNCH;
  const int nbm = (p.M + C::BM - 1) / C::BM, nbn = (p.N + C::BN - 1) / C::BN;
  hipLaunchKernelGGL((v3::gemm_w4a4_v3_kernel<C, false, true>), dim3((unsigned)(nbm * nbn), (unsigned)p.splits), dim3(C::NT),
                     C::LDS_BYTES, s, p);
  const int64_t MN = (int64_t)p.M * p.N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN / 8 + 255) / 256)), dim3(256), 0, s, p.ws, p.D, MN, p.splits);
  return check_launch();
}

template <class C, bool TRACE = false>
static int launch_v3_cfg(const GemmParams &p, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  if (ensure_max_lds(reinterpret_cast<const void *>(&v3::gemm_w4a4_v3_kernel<C, TRACE>), C::LDS_BYTES, attr_done) != ATOM_OK) return ATOM_ERR_LAUNCH;
  const int nbm = (p.M + C::BM - 1) / C::BM, nbn = (p.N + C::BN - 1) / C::BN;
  hipLaunchKernelGGL((v3::gemm_w4a4_v3_kernel<C, TRACE>), dim3((unsigned)(nbm * nbn)), dim3(C::NT), C::LDS_BYTES, s, p);
  return check_launch();
}

int launch_gemm_v3(const GemmParams &p, int cfg, hipStream_t s) {
  if (p.splits > 1 && p.ws) {
    if (cfg >= 20) return launch_v3_splitk<v3::Cfg<64, 128, 3, 2, true>>(p, s);
    return cfg == 4 ? launch_v3_splitk<v3::Cfg<64, 64, 3, 2>>(p, s) : launch_v3_splitk<v3::Cfg<64, 128, 3, 2>>(p, s);
  }
  switch (cfg) {
    case 1: return p.b_pairs ? launch_v3_cfg<v3::Cfg<256, 128, 3, 4, false, true>>(p, s)
                             : launch_v3_cfg<v3::Cfg<256, 128, 3>>(p, s);   // 4 waves, two workgroups per CU
    case 2: return launch_v3_cfg<v3::Cfg<128, 256, 3>>(p, s);   // 4 waves (1 x 4), two workgroups per CU
    case 3: return launch_v3_cfg<v3::Cfg<256, 128, 2>>(p, s);
    case 4: return launch_v3_cfg<v3::Cfg<64, 64, 3, 2>>(p, s);     // skinny M: one wave per workgroup, 64x64 tile
    case 5: return launch_v3_cfg<v3::Cfg<64, 128, 3, 2>>(p, s);    // skinny M: two waves, 64x128 tile
    case 6: return launch_v3_cfg<v3::Cfg<128, 64, 3, 4>>(p, s);    // one wave, 128x64 tile
#ifdef ATOM_TOOLS
    case 10: return launch_v3_cfg<v3::Cfg<256, 256, 4>, true>(p, s);   // traced (p.Dsz = u64 trace buffer)
    case 11: return launch_v3_cfg<v3::Cfg<256, 128, 3>, true>(p, s);
    case 30: return launch_v3_cfg<v3::Cfg<256, 256, 3, 4, true>, true>(p, s);   // traced
#endif
    case 20: return p.b_pairs ? launch_v3_cfg<v3::Cfg<256, 256, 3, 4, true, true>>(p, s)
                              : launch_v3_cfg<v3::Cfg<256, 256, 3, 4, true>>(p, s);   // wide activations (p.A4 = int8 [M, K4])
    case 24: return launch_v3_cfg<v3::Cfg<64, 64, 3, 2, true>>(p, s);
    case 25: return launch_v3_cfg<v3::Cfg<64, 128, 3, 2, true>>(p, s);
    default: return p.b_pairs ? launch_v3_cfg<v3::Cfg<256, 256, 4, 4, false, true>>(p, s)
                              : launch_v3_cfg<v3::Cfg<256, 256, 4>>(p, s);  // == v2 geometry
  }
}

}  // namespace atom
