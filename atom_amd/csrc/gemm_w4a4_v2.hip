// W4A4 GEMM, prefill kernel v2 for gfx950: "packed int4 through LDS-DMA, widen in registers".
//
// What the v0/v1 kernel (gemm_w4a4.hip) taught us on hardware (4096^3, ablations in profiles/r01_ablation.txt):
// skeleton (fragment ds_reads + epilogue) 41 us + MFMA 31 us + register->LDS stage stores 66 us + dequant
// VALU 48 us, and NONE of it overlapped (one lock-step barrier per K-group, int8-expanded LDS tiles whose
// ds_write/ds_read traffic alone nearly saturates the LDS pipe).  v2 therefore:
//
//  * moves the PACKED int4 operands HBM -> LDS with the DMA path (global_load_lds_dwordx4, 1 KiB per wave
//    instruction): no staging VGPRs, no ds_write, half the LDS bytes; an NS-deep ring of 33 KiB stages with
//    counted s_waitcnt vmcnt(N) + raw s_barrier keeps NS-2 stages in flight across every barrier;
//  * LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle is applied to the SOURCE address:
//    16-byte chunk j of row r lands in slot j ^ ((r>>2)&3); fragment ds_read_b128s use the same involution
//    (rows are 64 B, four rows per 256-B bank row: the XOR spreads a 16-lane read group over all 16 slots);
//  * 256x256 block tile, 8 waves = two per SIMD (VALU ops cannot read AGPRs, so the useful register budget is
//    256 VGPRs per wave whatever the occupancy; two waves per SIMD then overlap one wave's dequant VALU with the
//    other's MFMAs for free), wave tile 128(m) x 64(n): 0.375 ds_read_b128 per MFMA instead of 1.5;
//  * int4 -> int8 widening in registers ((v<<4)&0xF0F0F0F0, v&0xF0F0F0F0; even/odd k split on both operands);
//  * the integer accumulator of a 32x32 tile lives only for the 4 MFMAs of one K-group (one 16-register
//    temporary instead of a per-tile int32 accumulator), so the FP32 running sums (128 registers) fit; dequant is
//    3 VALU ops/element since round 5 (magic-number trick: idot = acc - 1.5 * 2^23 exactly; scale product; FMA -- rounds 1-4
//    folded the token scale into the magic fma, 2 ops/element, under the contract that rounded twice per group).
//
// Arithmetic contract (include/atom_hip.h): per int4 group  s = sA[m,g] * sB[g,n] (exact in FP32); c = fma(idot, s, c);
// the 128 INT8 keeper columns arrive as two 64-column stages and are multiplied in one step (one de-quantisation).
#include "common.h"

namespace atom {

namespace v2 {

constexpr float kMagic = 12582912.0f;
constexpr int kMagicBits = 0x4B400000;
constexpr int BM = 256, BN = 256, NT = 512;    // 8 waves = 2 per SIMD, <= 256 VGPRs each
constexpr int WGN = 4;                         // wave grid 2(m) x 4(n); wave tile 128(m) x 64(n)
constexpr int TM = 4, TN = 2;                  // 32x32 MFMA tiles per wave
constexpr int ROWS = BM + BN;                  // weights rows [0,256), activation rows [256,512)
constexpr int DATA_BYTES = ROWS * 64;          // packed: 64 B per row per K-group
constexpr int SB_OFF = DATA_BYTES;             // sB: 256 fp16, dense (dword DMA of channel pairs)
constexpr int SA_OFF = DATA_BYTES + 512;       // sA: 256 dwords, fp16 in the low half (ushort DMA, see probe)
constexpr int STAGE_BYTES = DATA_BYTES + 512 + 1024;
constexpr int GLDS_PER_STAGE = 5;              // per wave: 4 data + 1 scale DMA instruction

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// Per-lane DMA source offsets (bytes from the operand base), computed once: the K offset of a step is wave-uniform
// and goes into the scalar base, so issuing a stage costs no per-lane address arithmetic and 6 VGPRs in total.
struct StageAddr {
  unsigned data[4];      // row*stride + chunk*16 for the wave's 4 data DMA instructions (int4 geometry)
  unsigned data8[4];     // same for the keeper geometry (row stride 128)
  unsigned scale;        // element offset of this lane's scale (sA: row / replicated index; sB: channel pair)
};

__device__ __forceinline__ void make_stage_addr(const GemmParams &p, int wave, int lane, int m0, int n0, StageAddr &a) {
  const bool isW = wave < 4;                                  // waves 0-3 stream weights; 4-7 activations
  const int lim = isW ? p.N - 1 : p.M - 1;
  const int org = isW ? n0 + wave * 64 : m0 + (wave - 4) * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = i * 16 + (lane >> 2);                      // row inside this wave's 64
    const int row = wave * 64 + rl;                           // row inside the 512-row stage
    const int j = (lane & 3) ^ ((row >> 2) & 3);              // logical chunk that must land in slot lane&3
    const int idx = min(org + rl, lim);                       // clamp tails: results never stored
    a.data[i] = (unsigned)idx * (unsigned)p.K4h + j * 16;
    a.data8[i] = (unsigned)idx * kKeeper + j * 16;
  }
  if (wave < 4) {
    const int idx = min(m0 + wave * 64 + lane, p.M - 1);
    a.scale = p.ref_layout ? ref_scale_index(idx) : idx;
  } else {
    a.scale = min(n0 + (wave & 1) * 128 + 2 * lane, p.N - 2);
  }
}

// Issue the DMA of K-step `step` into ring slot `slot` (this wave's share: 64 rows + its scale slice).
__device__ __forceinline__ void issue_stage(const GemmParams &p, int step, char *slot, int wave, const StageAddr &a) {
  const bool int4 = step < p.G;
  const bool isW = wave < 4;
  const uint8_t *base = (isW ? (int4 ? p.B4 : p.B8) : (int4 ? p.A4 : p.A8)) + (int4 ? step : step - p.G) * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned off = int4 ? a.data[i] : a.data8[i];
    lds_dma<16>(base + off, slot + (wave * 64 + i * 16) * 64);
  }
  const bool keeper = step >= p.G;
  const int g = min(step, p.G - 1);
  if (wave < 4) {
    // sA: one fp16 per lane.  global_load_lds_ushort writes ONE ZERO-EXTENDED DWORD per lane (measured,
    // tools/probes/glds_probe.cpp) -> LDS image sA[m] at SA_OFF + 4*m.  Handles the replicated layout's gather.
    const half_t *sb = keeper ? p.sA8 : p.sA + (int64_t)g * p.ldA;
    lds_dma<2>(sb + a.scale, slot + SA_OFF + wave * 256);
  } else {
    // sB: a dword (two adjacent channels) per lane, dense fp16 image.  Waves 6,7 repeat waves 4,5 (same bytes to
    // the same place) so that every wave issues the same number of DMA instructions per stage.
    const half_t *sb = keeper ? p.sB8 : p.sB + (int64_t)g * p.N;
    lds_dma<4>(sb + a.scale, slot + SB_OFF + (wave & 1) * 256);
  }
}

__device__ __forceinline__ void widen(const v4u p, v4i &lo, v4i &hi) {
  const v4u l = (p << 4) & 0xF0F0F0F0u;       // even elements * 16
  const v4u h = p & 0xF0F0F0F0u;              // odd  elements * 16
  lo = __builtin_bit_cast(v4i, l);
  hi = __builtin_bit_cast(v4i, h);
}

// Read the fragments of one 32-row operand tile for one K-step.  INT4: 2 chunks -> 4 MFMA k-steps;
// keeper (raw int8, 64 columns per step): 2 chunks -> 2 MFMA k-steps.
// ABL (tuning only): 1 = no dequant VALU, 2 = no widening VALU, 4 = no MFMA, 8 = no DMA after the prologue,
// 16 = no fragment ds_reads
template <bool INT4, int ABL = 0>
__device__ __forceinline__ void load_frag(const char *slot, int row, int h, v4i (&f)[4]) {
  const int sw = (row >> 2) & 3;
  const char *rb = slot + row * 64;
  v4u c0, c1;
  if constexpr (ABL & 16) {
    c0 = v4u{(unsigned)row, (unsigned)h, 3u, 4u};
    c1 = v4u{(unsigned)row, 7u, (unsigned)h, 4u};
    asm volatile("" : "+v"(c0), "+v"(c1));
  } else {
    c0 = *reinterpret_cast<const v4u *>(rb + (((0 + h) ^ sw) << 4));
    c1 = *reinterpret_cast<const v4u *>(rb + (((2 + h) ^ sw) << 4));
  }
  if constexpr (INT4 && (ABL & 2)) {
    f[0] = __builtin_bit_cast(v4i, c0); f[1] = f[0]; f[2] = __builtin_bit_cast(v4i, c1); f[3] = f[2];
  } else if constexpr (INT4) {
    widen(c0, f[0], f[1]);
    widen(c1, f[2], f[3]);
  } else {
    f[0] = __builtin_bit_cast(v4i, c0);
    f[1] = __builtin_bit_cast(v4i, c1);
  }
}

template <bool INT4, int ABL = 0>
__device__ __forceinline__ void compute_step(const char *slot, int wm, int wn, int lane, float (&c)[TN][TM][16]) {
  constexpr int KS = INT4 ? 4 : 2;
  const int l31 = lane & 31, h = lane >> 5;

  v16i magic;
#pragma unroll
  for (int i = 0; i < 16; ++i) magic[i] = kMagicBits;

  // weight fragments (2 n-tiles) stay resident for the whole step
  v4i af[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) load_frag<INT4, ABL>(slot, wn * 64 + tn * 32 + l31, h, af[tn]);

  // software pipeline over the 4 m-tiles: the two packed 16-byte chunks (8 VGPRs) and the scale of tile tm+1 are
  // requested from LDS before tile tm's MFMAs/dequant, so their latency is covered by work instead of a stall.
  v4u pk0, pk1;
  half_t sah;
  auto request = [&](int tm) {
    const int row = 256 + wm * 128 + tm * 32 + l31;
    const int sw = (row >> 2) & 3;
    const char *rb = slot + row * 64;
    pk0 = *reinterpret_cast<const v4u *>(rb + (((0 + h) ^ sw) << 4));
    pk1 = *reinterpret_cast<const v4u *>(rb + (((2 + h) ^ sw) << 4));
    sah = *reinterpret_cast<const half_t *>(slot + SA_OFF + (wm * 128 + tm * 32 + l31) * 4);
  };
  request(0);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    __builtin_amdgcn_sched_barrier(0);
    v4i bf[4];
    if constexpr (INT4 && !(ABL & 2)) {
      widen(pk0, bf[0], bf[1]);
      widen(pk1, bf[2], bf[3]);
    } else {
      bf[0] = __builtin_bit_cast(v4i, pk0); bf[1] = __builtin_bit_cast(v4i, pk1); bf[2] = bf[0]; bf[3] = bf[1];
    }
    // int4 operands are widened to 16*value on both sides: fold 1/256 into the activation scale (exact)
    const float sa = (float)sah * (INT4 ? (1.0f / 256.0f) : 1.0f);
    if (tm + 1 < TM) request(tm + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      // one tile at a time: 4 chained MFMAs, then its dequant.  The fence keeps the scheduler from interleaving the
      // two n-tiles (which would need a second accumulator + 32 temporaries and spills the running sums); the
      // co-resident wave on this SIMD runs its MFMAs under this wave's dequant instead.
      __builtin_amdgcn_sched_barrier(0);
      v16i a;
      if constexpr (ABL & 4) {
        a = magic;
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(a) : "v"(af[tn][s]), "v"(bf[s]));
      } else {
        a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][0], bf[0], magic, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < KS; ++s) a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][s], bf[s], a, 0, 0, 0);
      }
      if constexpr (ABL & 1) {
        asm volatile("" ::"v"(a));
        continue;
      }
      // the 16 weight scales of this tile's accumulator registers: n = nl + 8q + 4h + {0..3} (re-read per tile: 8
      // transient registers instead of 16 resident ones)
      v2u sbp[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        sbp[q] = *reinterpret_cast<const v2u *>(slot + SB_OFF + (wn * 64 + tn * 32 + 8 * q + 4 * h) * 2);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const half_t *hv = reinterpret_cast<const half_t *>(&sbp[r >> 2]);
        const float idot = __int_as_float(a[r]) - kMagic;       // exact (the register read as a float is 12582912 + idot)
        c[tn][tm][r] = __builtin_fmaf(idot, (float)hv[r & 3] * sa, c[tn][tm][r]);   // the contract: exact scale product, one rounding
        // pin the update HERE: otherwise LLVM defers all 128 second-stage FMAs of a step to its end and keeps 128
        // temporaries alive (measured: 180 spilled VGPRs)
        asm volatile("" : "+v"(c[tn][tm][r]));
      }
    }
  }
}

// The keeper: its 128 INT8 columns arrive as TWO stages of 64-byte rows (slot0 = columns 0..63, slot1 = 64..127) and are multiplied
// in ONE step: four chained MFMAs per 32x32 tile into one accumulator, one de-quantisation -- as the reference kernel does
// (Dense_layer_gemm_i4_o16.cuh:640-691; rounds 1-2 of this repository de-quantised the two halves separately).
__device__ __forceinline__ void compute_keeper(const char *slot0, const char *slot1, int wm, int wn, int lane, float (&c)[TN][TM][16]) {
  const int l31 = lane & 31, h = lane >> 5;
  v16i magic;
#pragma unroll
  for (int i = 0; i < 16; ++i) magic[i] = kMagicBits;
  v4i af[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    v4i f0[4], f1[4];
    load_frag<false, 0>(slot0, wn * 64 + tn * 32 + l31, h, f0);
    load_frag<false, 0>(slot1, wn * 64 + tn * 32 + l31, h, f1);
    af[tn][0] = f0[0]; af[tn][1] = f0[1]; af[tn][2] = f1[0]; af[tn][3] = f1[1];
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    __builtin_amdgcn_sched_barrier(0);
    const int ml = wm * 128 + tm * 32 + l31;
    v4i b0[4], b1[4];
    load_frag<false, 0>(slot0, 256 + ml, h, b0);
    load_frag<false, 0>(slot1, 256 + ml, h, b1);
    const float sa = (float)*reinterpret_cast<const half_t *>(slot0 + SA_OFF + ml * 4);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      __builtin_amdgcn_sched_barrier(0);
      v16i a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][0], b0[0], magic, 0, 0, 0);
      a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][1], b0[1], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][2], b1[0], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][3], b1[1], a, 0, 0, 0);
      v2u sbp[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        sbp[q] = *reinterpret_cast<const v2u *>(slot0 + SB_OFF + (wn * 64 + tn * 32 + 8 * q + 4 * h) * 2);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const half_t *hv = reinterpret_cast<const half_t *>(&sbp[r >> 2]);
        const float idot = __int_as_float(a[r]) - kMagic;       // exact (the register read as a float is 12582912 + idot)
        c[tn][tm][r] = __builtin_fmaf(idot, (float)hv[r & 3] * sa, c[tn][tm][r]);   // the contract: exact scale product, one rounding
        asm volatile("" : "+v"(c[tn][tm][r]));
      }
    }
  }
}

// Ping-pong variant (ABL bit 64, experimental): the MFMA chain of tile i+1 is issued BEFORE the dequant of tile i, so
// the 12 wait states after a chain and the dequant's dependency latency are covered by matrix work of the same wave.
template <bool INT4>
__device__ __forceinline__ void compute_step_pp(const char *slot, int wm, int wn, int lane, float (&c)[TN][TM][16]) {
  constexpr int KS = INT4 ? 4 : 2;
  const int l31 = lane & 31, h = lane >> 5;
  v16i magic;
#pragma unroll
  for (int i = 0; i < 16; ++i) magic[i] = kMagicBits;
  v4i af[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) load_frag<INT4, 0>(slot, wn * 64 + tn * 32 + l31, h, af[tn]);

  auto chain = [&](const v4i (&bf)[4], int tn) {
    v16i a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][0], bf[0], magic, 0, 0, 0);
#pragma unroll
    for (int s = 1; s < KS; ++s) a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn][s], bf[s], a, 0, 0, 0);
    return a;
  };
  auto dequant = [&](const v16i &a, int tn, int tm, float sa) {
    v2u sbp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      sbp[q] = *reinterpret_cast<const v2u *>(slot + SB_OFF + (wn * 64 + tn * 32 + 8 * q + 4 * h) * 2);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const half_t *hv = reinterpret_cast<const half_t *>(&sbp[r >> 2]);
      const float idot = __int_as_float(a[r]) - kMagic;
      c[tn][tm][r] = __builtin_fmaf(idot, (float)hv[r & 3] * sa, c[tn][tm][r]);
      asm volatile("" : "+v"(c[tn][tm][r]));
    }
  };

  v16i a_prev;
  float sa_prev = 0.f;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int ml = wm * 128 + tm * 32 + l31;
    v4i bf[4];
    load_frag<INT4, 0>(slot, 256 + ml, h, bf);
    const float sa = (float)*reinterpret_cast<const half_t *>(slot + SA_OFF + ml * 4) * (INT4 ? (1.0f / 256.0f) : 1.0f);
    __builtin_amdgcn_sched_barrier(0);
    v16i a0 = chain(bf, 0);                       // tile (tm, 0) in flight ...
    __builtin_amdgcn_sched_barrier(0);
    if (tm > 0) dequant(a_prev, 1, tm - 1, sa_prev);     // ... while tile (tm-1, 1) is dequantised
    __builtin_amdgcn_sched_barrier(0);
    v16i a1 = chain(bf, 1);                       // tile (tm, 1) in flight ...
    __builtin_amdgcn_sched_barrier(0);
    dequant(a0, 0, tm, sa);                  // ... while tile (tm, 0) is dequantised
    __builtin_amdgcn_sched_barrier(0);
    a_prev = a1; sa_prev = sa;
  }
  dequant(a_prev, 1, TM - 1, sa_prev);
}

template <int NS, int ABL = 0, bool O4 = false>
__global__ __launch_bounds__(NT) void gemm_w4a4_v2_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // Block -> tile.  Workgroup b runs on XCD b % 8 (observed; speed only, never correctness): give every XCD a
  // contiguous chunk of the tile sequence (bijective for any grid size), and walk the tiles in bands of 4 m-tiles so
  // that the blocks resident on one XCD at a time share A/B panels in that XCD's private L2.
  const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
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
  const int m0 = bm * BM, n0 = bn * BN;

  float c[TN][TM][16];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[a][b][r] = 0.f;

  const int nsteps = p.G + 2;                 // DMA stages: G int4 groups + the two 64-column halves of the INT8 keeper (ONE compute step)
  // prologue: NS-1 stages in flight.  Steps past the end re-load the last step into a slot nobody reads
  // again, so that the vmcnt bookkeeping below is the same constant on every iteration.
  StageAddr sa_;
  make_stage_addr(p, wave, lane, m0, n0, sa_);
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue_stage(p, min(s, nsteps - 1), lds + s * STAGE_BYTES, wave, sa_);

  // Two loops (int4 groups, then the two keeper halves) instead of one loop with a branch: with both bodies in one
  // loop the register allocator parks half of the running sums in scratch (measured: 120-180 spilled VGPRs).
#define ATOM_V2_STEP(INT4)                                                                                         \
  {                                                                                                                \
    /* my own DMA parts of stage `step` have landed once at most NS-2 younger stages are outstanding */            \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ABL & 8) ? 0 : GLDS_PER_STAGE * (NS - 2)) : "memory");               \
    if (!(ABL & 32)) __builtin_amdgcn_s_barrier(); /* everyone's parts landed; everyone finished step-1 */         \
    /* refill the slot consumed in step-1 */                                                                       \
    if (!(ABL & 8))                                                                                                \
      issue_stage(p, min(step + NS - 1, nsteps - 1), lds + ((step + NS - 1) % NS) * STAGE_BYTES, wave, sa_);       \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    if constexpr (ABL & 64)                                                                                        \
      compute_step_pp<INT4>(lds + (step % NS) * STAGE_BYTES, wm, wn, lane, c);                                     \
    else                                                                                                           \
      compute_step<INT4, ABL>(lds + (step % NS) * STAGE_BYTES, wm, wn, lane, c);                                   \
  }
  if constexpr (ABL & 128) {   // experiment: static priority for the younger half of the workgroup
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  }
  if constexpr (ABL & 256) {   // experiment: start the two waves of a SIMD half a tile apart
    if (wave >= 4) __builtin_amdgcn_s_sleep(4);
  }
  int step = 0;
  for (; step < p.G; ++step) ATOM_V2_STEP(true)
#undef ATOM_V2_STEP
  {                                           // the keeper: both halves (stages G, G + 1) in one compute step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (NS == 2) {                  // (two-stage ring: the second half only fits now that stage G - 1 has been read)
      issue_stage(p, p.G + 1, lds + ((p.G + 1) % NS) * STAGE_BYTES, wave, sa_);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    compute_keeper(lds + (p.G % NS) * STAGE_BYTES, lds + ((p.G + 1) % NS) * STAGE_BYTES, wm, wn, lane, c);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the dummy tail DMAs: they still target the LDS ring
  __builtin_amdgcn_s_barrier();                       // nobody reads stage data any more

  if constexpr (O4) {
    // u4 epilogue (reference: e2e/punica-atom/punica/ops/csrc/GEMM/DenseLayerGEMM_i4_o4.cu:704-788): every 128-wide
    // output group of a row is asymmetrically quantised from the FP32 accumulators:
    //   scale = (max-min)/15, zero = -min, q = clamp(round_half_away((x+zero)*(1/scale)), 0, 15)
    // (the reference's local_max_min takes abs() of BOTH extrema, :73-80 -- a bug that only cancels for non-negative
    // tiles; we implement the intended min/max, which is what its consumer de-quantises: q*scale - zero,
    // kernels/include/flashinfer/quantization.cuh:59-84).
    const int l31 = lane & 31, h = lane >> 5;
    float *mm = reinterpret_cast<float *>(lds);                 // [wave][128 rows][2] = 8 KB
    char *ep = lds + 8192 + wave * (128 * 48);                  // per-wave [128 rows][32 B], stride 48
    float mn[TM], mx[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      float lo = p.o4_ref ? fabsf(c[0][tm][0]) : c[0][tm][0], hi = lo;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = p.o4_ref ? fabsf(c[tn][tm][r]) : c[tn][tm][r];   // (reference code: extrema of |x|, see atom_hip.h)
          lo = fminf(lo, v);
          hi = fmaxf(hi, v);
        }
      lo = fminf(lo, __shfl_xor(lo, 32));                       // the other half-wave holds the other 32 features
      hi = fmaxf(hi, __shfl_xor(hi, 32));
      mn[tm] = lo; mx[tm] = hi;
      if (h == 0) {
        mm[(wave * 128 + tm * 32 + l31) * 2 + 0] = lo;
        mm[(wave * 128 + tm * 32 + l31) * 2 + 1] = hi;
      }
    }
    __syncthreads();
    const int partner = wave ^ 1;                               // the wave holding the other 64 columns of the group
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const float lo = fminf(mn[tm], mm[(partner * 128 + tm * 32 + l31) * 2 + 0]);
      const float hi = fmaxf(mx[tm], mm[(partner * 128 + tm * 32 + l31) * 2 + 1]);
      const float scale = (hi - lo) / 15.f;
      const float zero = -lo;
      const float rs = 1.0f / scale;
      const int m = m0 + wm * 128 + tm * 32 + l31;
      if (h == 0 && (wn & 1) == 0 && m < p.M) {
        const int grp = (n0 + wn * 64) >> 7;
        if (grp * 128 < p.N) {
          half_t *dst = p.Dsz + ((int64_t)m * (p.N >> 7) + grp) * 2;
          dst[0] = f2h(scale);
          dst[1] = f2h(zero);
        }
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned w = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            w |= (p.o4_ref ? o4_code<true>(c[tn][tm][4 * q + k], zero, rs, scale) : o4_code<false>(c[tn][tm][4 * q + k], zero, rs, scale))
                 << (4 * k);
          *reinterpret_cast<unsigned short *>(ep + (tm * 32 + l31) * 48 + (tn * 32 + 8 * q + 4 * h) / 2) = (unsigned short)w;
        }
    }
    // wave-private transpose region: two 16-byte chunks per row -> 16-byte global stores, 32 rows per instruction
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = i * 32 + (lane >> 1), ch = lane & 1;
      const v4u v = *reinterpret_cast<const v4u *>(ep + rl * 48 + ch * 16);
      const int m = m0 + wm * 128 + rl;
      const int n = n0 + wn * 64 + ch * 32;
      if (m < p.M && n < p.N) *reinterpret_cast<v4u *>(p.D4 + ((int64_t)m * p.N + n) / 2) = v;
    }
    return;
  }

  // Epilogue.  A lane owns token m and 4 consecutive features per (tile, q): written straight to HBM that is an
  // 8-byte store per lane at a row stride -- 64 different cache lines per instruction (measured: 32 us of a 109 us
  // kernel).  Instead each wave transposes its 128x64 fp16 tile through its own LDS region (row stride 144 B: 16-byte
  // aligned for ds_read_b128, 2-way-only bank conflicts for the ds_write_b64) and stores full 128-byte row segments,
  // 16 bytes per lane.  Two passes of 64 rows keep the footprint inside the DMA ring.
  constexpr int EP_STRIDE = 144;
  char *ep = lds + wave * (64 * EP_STRIDE);
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
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
    // wave-private region: program order + the data dependence of the loads below is all the ordering needed
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rl = i * 8 + (lane >> 3);                 // row inside this half (0..63)
      const int ch = lane & 7;                            // 16-byte chunk inside the 128-byte row
      const v4u v = *reinterpret_cast<const v4u *>(ep + rl * EP_STRIDE + ch * 16);
      const int m = m0 + wm * 128 + half * 64 + rl;
      const int n = n0 + wn * 64 + ch * 8;
      if (m < p.M && n < p.N) *reinterpret_cast<v4u *>(p.D + (int64_t)m * p.N + n) = v;
    }
  }
}

}  // namespace v2

template <int NS, int ABL = 0, bool O4 = false>
static int launch_v2(const GemmParams &p, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  constexpr int lds_bytes = NS * v2::STAGE_BYTES > 8 * 64 * 144 ? NS * v2::STAGE_BYTES : 8 * 64 * 144;
  if (ensure_max_lds(reinterpret_cast<const void *>(&v2::gemm_w4a4_v2_kernel<NS, ABL, O4>), lds_bytes, attr_done) != ATOM_OK) return ATOM_ERR_LAUNCH;
  const int nbm = (p.M + v2::BM - 1) / v2::BM, nbn = (p.N + v2::BN - 1) / v2::BN;
  hipLaunchKernelGGL((v2::gemm_w4a4_v2_kernel<NS, ABL, O4>), dim3((unsigned)(nbm * nbn)), dim3(v2::NT), lds_bytes, s, p);
  return check_launch();
}

int launch_gemm_v2_o4(const GemmParams &p, hipStream_t s) { return launch_v2<4, 0, true>(p, s); }

int launch_gemm_v2(const GemmParams &p, int ns, hipStream_t s) {
  switch (ns) {
    case 3: return launch_v2<3>(p, s);
#ifdef ATOM_TOOLS   // ablation masks (profiles/r01_ablation_v1_v2.txt): tools build only
    case 1001: return launch_v2<4, 1>(p, s);
    case 1002: return launch_v2<4, 2>(p, s);
    case 1003: return launch_v2<4, 3>(p, s);
    case 1004: return launch_v2<4, 4>(p, s);
    case 1008: return launch_v2<4, 8>(p, s);
    case 1016: return launch_v2<4, 16>(p, s);
    case 1019: return launch_v2<4, 19>(p, s);
    case 1023: return launch_v2<4, 23>(p, s);
    case 1031: return launch_v2<4, 31>(p, s);
    case 1032: return launch_v2<4, 32>(p, s);
    case 1033: return launch_v2<4, 33>(p, s);
    case 1035: return launch_v2<4, 35>(p, s);
    case 1064: return launch_v2<4, 64>(p, s);
    case 1128: return launch_v2<4, 128>(p, s);
    case 1256: return launch_v2<4, 256>(p, s);
#endif
    default: return launch_v2<4>(p, s);
  }
}

}  // namespace atom
