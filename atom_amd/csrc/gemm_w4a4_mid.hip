// W4A4 GEMM for mid-size batches (17 .. ~1024 tokens) in the REFERENCE operand format (packed INT4 nibbles, fp16 scales): the kernel
// between the decode-batch kernel (gemm_w4a4_skinny.hip: one workgroup per 16 features, every workgroup pulls the whole activation
// matrix -- 15 us at 256 x 4096 x 4096, two thirds of its waves' life parked) and the 256-row tiles of the prefill kernels (one
// latency-bound tile per CU at these sizes).  The reference runs ONE tile kernel over every batch size
// (kernels/src/GEMM/bench_dense_layer_gemm_i4_o16.cu:64-69: bsz 16 .. 4096 through the same 128 x 128 cp.async tile); this is that
// role, designed for what bounds a mid-size batch on gfx950:
//   * tiles of 64 tokens x 64 features over the WHOLE K range: M / 64 x N / 64 workgroups (256 at 256 x 4096: one per CU), each
//     weight byte staged (M / 64) times and each activation byte (N / 64) times out of L2 -- 69 MB through the LDS-DMA path at
//     256 x 4096 x 4096 against 138 MB of per-lane global loads in the decode-batch kernel -- and no exchange of partial sums:
//     every output element is ONE ordered sum over the K steps, the tile kernels' order (include/atom_hip.h), bit for bit;
//   * a DEEP LDS ring: a K step's stage is 8.5 KB (64 + 64 rows x 64 B + scales), far less than what one CU must keep in flight to
//     cover the memory latency (~48 KB, MI355X_MICROARCH.md), and a 3-stage ring makes every step wait for a round trip (the
//     ~3,000 cycles per K step of the round-2..4 kernels at these sizes).  16 stages (136 KB, one workgroup per CU) or 8 (two
//     workgroups per CU when the shape has more than one tile per CU) are requested before the first step; a step waits for a
//     stage issued NS - 2 steps earlier with a COUNTED s_waitcnt, so the wait is a formality and the loop runs at its issue rate;
//   * operands straight from the packed format: LDS-DMA (1 KiB per wave instruction, no VGPR round trip) of 16-row pieces whose
//     rows are picked per lane so that the LDS image is already in MFMA order -- fragment (h, k) row i = feature 32 h + 2 i + k, so a
//     lane ends with 8 consecutive features (one 16-byte store) and, with weight_channel_group = 2, the two micro-tiles of a pair
//     share their scale products (ATOM_B_SCALE_PAIRS) -- with the 16-byte chunks of a row XOR-swizzled by (i >> 1) & 3: conflict-free
//     ds_read_b128 fragments; a packed chunk's 32 codes become the two INT8 operands (x << 4) & 0xF0F0F0F0 / x & 0xF0F0F0F0
//     (= 16 * code, the same k-permutation on both sides: two v_mfma_i32_16x16x64_i8 give 256 * idot exactly; 1 / 256 rides in
//     the token scale);
//   * the MFMAs of step s are consumed in step s + 1, behind its LDS loads: no MFMA -> VALU result stall, and the de-quantisation
//     covers the fragment loads' latency.
// Arithmetic: the contract of include/atom_hip.h -- s = sA * sB exact, c = fma(idot, s, c) per group in order, the keeper as ONE
// 128-column dot product de-quantised once -- so results are bit-identical to the tile kernels (gemm_w4a4_v3.hip, gemm_w4a4_f6.hip
// with atom_gemm_w4a4_f6_order == 1) and to oracle/atom_oracle.c:oracle_gemm_w4a4_f16.
#include <type_traits>
#include "common.h"

namespace atom {
namespace mid {

constexpr int BM = 64, BN = 64;
constexpr int W_OFF = 0;          // 4 fragments x 1 KiB: fragment f = 2 h + k, row i = feature 32 h + 2 i + k of the tile
constexpr int A_OFF = 4096;       // 4 token blocks x 1 KiB: row i = token 16 b + i
constexpr int SB_OFF = 8192;      // 64 weight scales fp16 (a dword piece moves 128: the upper half is not read)
constexpr int SA_OFF = 8448;      // 64 token scales, fp16 zero-extended to a dword each (ushort pieces)
constexpr int STAGE = 8704;

template <int NW_, int NS_, bool PAIR_>
struct Cfg {
  static constexpr int NW = NW_, NS = NS_, NT = NW * 64;
  static constexpr bool PAIR = PAIR_;
  static constexpr int TBW = 8 / NW;     // token blocks per wave: wave = (token-block group w >> 1, feature half w & 1)
  static constexpr int DPW = 8 / NW;     // 1 KiB data pieces per wave and stage
  static constexpr int PPW = DPW + 1;    // waves 0 / 1: + the weight / token scale piece
  static constexpr int LDS_BYTES = NS * STAGE;
  static_assert(NW == 4 || NW == 8, "waves per workgroup");
  static_assert(NS >= 3 && PPW * (NS - 2) < 64 && LDS_BYTES <= 160 * 1024, "ring depth");
};

template <class C>
struct Dma {               // loop invariant: per lane the offsets, per wave (SGPRs) the operand it stages
  unsigned voff[C::DPW];   // int4 stages: row * K4h + 16 * source chunk (from B4 / A4 + 64 g)
  unsigned kvoff[C::DPW];  // keeper halves: row * 128 + 16 * source chunk (from B8 / A8 + 64 half)
  unsigned svoff;          // scale piece: byte offset from the group's scale row (weight: feature pair; token: its index in the layout)
  const uint8_t *d4, *d8;  // the wave's data operand: weights (waves whose pieces are 0..3) or activations
  const half_t *s4, *s8;   // its scale operand: weight scales (even waves) or token scales (odd waves)
  int64_t sstride;         // halves between the groups of s4
  int lds0;                // LDS byte offset of the wave's first piece inside a stage
};

template <class C>
__device__ __forceinline__ void make_dma(const GemmParams &p, int wave, int lane, int m0, int n0, Dma<C> &d) {
  const int i = lane >> 2, j = lane & 3;
  const unsigned chunk = (unsigned)((j ^ ((i >> 1) & 3)) << 4);    // LDS slot j of row i receives source chunk j ^ ((i >> 1) & 3)
  const bool is_w = wave * C::DPW < 4;                             // pieces 0..3: weight fragments, 4..7: token blocks
#pragma unroll
  for (int k = 0; k < C::DPW; ++k) {
    const int piece = wave * C::DPW + k;
    const unsigned row = is_w ? (unsigned)(n0 + 32 * (piece >> 1) + 2 * i + (piece & 1)) : (unsigned)min(m0 + 16 * (piece - 4) + i, p.M - 1);
    d.voff[k] = row * (unsigned)p.K4h + chunk;
    d.kvoff[k] = row * (unsigned)kKeeper + chunk;
  }
  d.lds0 = wave * C::DPW * 1024;
  d.d4 = is_w ? p.B4 : p.A4;
  d.d8 = is_w ? p.B8 : p.A8;
  if (wave & 1) {
    const int m = min(m0 + lane, p.M - 1);
    d.svoff = (unsigned)(p.ref_layout ? ref_scale_index(m) : m) * 2u;
    d.s4 = p.sA; d.s8 = p.sA8; d.sstride = p.ldA;
  } else {
    d.svoff = (unsigned)min(n0 + 2 * lane, p.N - 2) * 2u;
    d.s4 = p.sB; d.s8 = p.sB8; d.sstride = p.N;
  }
}

// stage `st`: 0 .. G - 1 the int4 groups, G / G + 1 the two 64-column halves of the keeper (same row layout, INT8 bytes)
template <class C>
__device__ __forceinline__ void issue_stage(const GemmParams &p, int st, unsigned slot, int wave, const Dma<C> &d) {
  const bool k8 = st >= p.G;
  const uint8_t *base = k8 ? d.d8 + (st - p.G) * 64 : d.d4 + (int64_t)st * 64;
  const half_t *sc = k8 ? d.s8 : d.s4 + (int64_t)st * d.sstride;
#pragma unroll
  for (int k = 0; k < C::DPW; ++k) lds_dma_sv<16>(base, k8 ? d.kvoff[k] : d.voff[k], slot + d.lds0 + k * 1024);
  if (wave == 1) lds_dma_sv<2>(sc, d.svoff, slot + SA_OFF);       // one scale piece each on waves 0 and 1 only: an LDS-DMA instruction costs
  else if (wave == 0) lds_dma_sv<4>(sc, d.svoff, slot + SB_OFF);   // its issuer ~100 cycles whatever it moves (their s_waitcnt counts differ)
}

__device__ __forceinline__ v4i even_codes(v4u x) {       // low nibbles  -> int8 16 * code
  return v4i{(int)((x.x << 4) & 0xF0F0F0F0u), (int)((x.y << 4) & 0xF0F0F0F0u), (int)((x.z << 4) & 0xF0F0F0F0u),
             (int)((x.w << 4) & 0xF0F0F0F0u)};
}
__device__ __forceinline__ v4i odd_codes(v4u x) {        // high nibbles -> int8 16 * code
  return v4i{(int)(x.x & 0xF0F0F0F0u), (int)(x.y & 0xF0F0F0F0u), (int)(x.z & 0xF0F0F0F0u), (int)(x.w & 0xF0F0F0F0u)};
}

// what a step leaves for the next one to de-quantise: the integer dot products and the scales they go with
template <class C>
struct Pending {
  v4i acc[C::TBW][2];
  float sa[C::TBW];        // token scale (x 1 / 256 for the widened int4 operands)
  float sb[8];             // weight scales of the lane's 8 features, [2 r + k]
};

// c[k][t][r] = fma(idot, sa * sb, c): lane = token 16 (tb0 + t) + l15, feature 32 h + 8 kb + 2 r + k (the contract; PAIR: channels 2 r,
// 2 r + 1 share their scale, one product for both micro-tiles)
template <class C, bool PAIR>
__device__ __forceinline__ void dequant(const Pending<C> &q, float (&c)[2][C::TBW][4]) {
#pragma unroll
  for (int t = 0; t < C::TBW; ++t) {
    float s[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[2 * r] = q.sa[t] * q.sb[2 * r];
      s[2 * r + 1] = PAIR ? s[2 * r] : q.sa[t] * q.sb[2 * r + 1];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c[k][t][r] = __builtin_fmaf((float)q.acc[t][k][r], s[2 * r + k], c[k][t][r]);
        asm volatile("" : "+v"(c[k][t][r]));
      }
  }
}

template <class C>
__global__ __launch_bounds__(C::NT) void gemm_w4a4_mid_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NS = C::NS, TBW = C::TBW;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < C::NW);
  const int h = wave & 1, tb0 = (wave >> 1) * TBW;
  const int nbm = (p.M + BM - 1) / BM, nbn = p.N / BN;
  int id = blockIdx.x;
  {                                                        // workgroup b runs on XCD b % 8: every XCD takes a contiguous run of tiles,
    const int nwg = nbm * nbn;                             // token tiles fastest -- the workgroups that share a weight tile share an L2
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m0 = (id % nbm) * BM, n0 = (id / nbm) * BN;

  Dma<C> d;
  make_dma<C>(p, wave, lane, m0, n0, d);
  const unsigned lds0 = lds_addr(lds);
  const int G = p.G, T = G + 2;                            // stages: G int4 groups + the keeper's two halves
#pragma unroll 1
  for (int s = 0; s < NS - 1; ++s) issue_stage<C>(p, min(s, T - 1), lds0 + s * STAGE, wave, d);

  const int l15 = lane & 15, kb = lane >> 4;
  const int coff = l15 * 64 + ((kb ^ ((l15 >> 1) & 3)) << 4);      // the lane's 16-byte chunk of a fragment row (swizzled)
  const int aw = W_OFF + 2 * h * 1024 + coff, aa = A_OFF + tb0 * 1024 + coff;
  const int asa = SA_OFF + (tb0 * 16 + l15) * 4, asb = SB_OFF + (32 * h + 8 * kb) * 2;

  float c[2][TBW][4];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int t = 0; t < TBW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[k][t][r] = 0.f;
  Pending<C> q;
#pragma unroll
  for (int t = 0; t < TBW; ++t) {
    q.sa[t] = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) q.acc[t][k] = v4i{0, 0, 0, 0};
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) q.sb[j] = 0.f;

  int slot = 0, dslot = NS - 1;                            // ring positions of the stage computed / requested in this step
  for (int step = 0; step < G; ++step) {
    if (wave < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PPW * (NS - 2)) : "memory");   // this wave's pieces of stage `step` have landed
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::DPW * (NS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();                          // ... everybody's; and everybody is done reading the slot of step - 1
    issue_stage<C>(p, min(step + NS - 1, T - 1), lds0 + dslot * STAGE, wave, d);   // (past the end: repeats into a dead slot keep the count)
    __builtin_amdgcn_sched_barrier(0);
    const char *sl = lds + slot * STAGE;
    v4u wf[2], bf[TBW];
    unsigned sah[TBW];
    wf[0] = *reinterpret_cast<const v4u *>(sl + aw);
    wf[1] = *reinterpret_cast<const v4u *>(sl + aw + 1024);
#pragma unroll
    for (int t = 0; t < TBW; ++t) {
      bf[t] = *reinterpret_cast<const v4u *>(sl + aa + t * 1024);
      sah[t] = *reinterpret_cast<const unsigned *>(sl + asa + t * 64);
    }
    const v4u sbv = *reinterpret_cast<const v4u *>(sl + asb);
    __builtin_amdgcn_sched_barrier(0);
    dequant<C, C::PAIR>(q, c);                             // the previous step's products, while this step's fragments arrive
    __builtin_amdgcn_sched_barrier(0);
    v4i we[2], wo[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) { we[k] = even_codes(wf[k]); wo[k] = odd_codes(wf[k]); }
#pragma unroll
    for (int t = 0; t < TBW; ++t) {
      const v4i be = even_codes(bf[t]), bo = odd_codes(bf[t]);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        v4i a = __builtin_amdgcn_mfma_i32_16x16x64_i8(we[k], be, v4i{0, 0, 0, 0}, 0, 0, 0);
        q.acc[t][k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wo[k], bo, a, 0, 0, 0);
      }
      q.sa[t] = (float)__builtin_bit_cast(half_t, (unsigned short)sah[t]) * (1.0f / 256.0f);
    }
    {
      const half_t *hv = reinterpret_cast<const half_t *>(&sbv);
#pragma unroll
      for (int j = 0; j < 8; ++j) q.sb[j] = (float)hv[j];
    }
    slot = slot + 1 == NS ? 0 : slot + 1;
    dslot = dslot + 1 == NS ? 0 : dslot + 1;
  }
  dequant<C, C::PAIR>(q, c);                               // the last int4 step
  // the keeper: stages G and G + 1 are its two 64-column halves (INT8 bytes in the int4 stage layout): ONE dot product per output,
  // two chained MFMAs, one de-quantisation with the per-channel keeper scales (never pair-shared: model/qLinearLayer.py:59)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    const char *s0 = lds + slot * STAGE, *s1 = lds + (slot + 1 == NS ? 0 : slot + 1) * STAGE;
    v4i w0[2], w1[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      w0[k] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(s0 + aw + k * 1024));
      w1[k] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(s1 + aw + k * 1024));
    }
    const v4u sbv = *reinterpret_cast<const v4u *>(s0 + asb);
    const half_t *hv = reinterpret_cast<const half_t *>(&sbv);
#pragma unroll
    for (int j = 0; j < 8; ++j) q.sb[j] = (float)hv[j];
#pragma unroll
    for (int t = 0; t < TBW; ++t) {
      const v4i b0 = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(s0 + aa + t * 1024));
      const v4i b1 = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(s1 + aa + t * 1024));
      const unsigned sah = *reinterpret_cast<const unsigned *>(s0 + asa + t * 64);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        v4i a = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0[k], b0, v4i{0, 0, 0, 0}, 0, 0, 0);
        q.acc[t][k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1[k], b1, a, 0, 0, 0);
      }
      q.sa[t] = (float)__builtin_bit_cast(half_t, (unsigned short)sah);
    }
    dequant<C, false>(q, c);
  }
  // a lane holds 8 consecutive features of one token per block: one 16-byte store each
#pragma unroll
  for (int t = 0; t < TBW; ++t) {
    const int m = m0 + 16 * (tb0 + t) + l15;
    if (m >= p.M) continue;
    v4u o;
    half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ov[2 * r] = f2h(c[0][t][r]);
      ov[2 * r + 1] = f2h(c[1][t][r]);
    }
    *reinterpret_cast<v4u *>(p.D + (int64_t)m * p.N + n0 + 32 * h + 8 * kb) = o;
  }
}

template <class C>
static int launch(const GemmParams &p, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  if (ensure_max_lds(reinterpret_cast<const void *>(&gemm_w4a4_mid_kernel<C>), C::LDS_BYTES, attr_done) != ATOM_OK) return ATOM_ERR_LAUNCH;
  const int nbm = (p.M + BM - 1) / BM, nbn = p.N / BN;
  hipLaunchKernelGGL((gemm_w4a4_mid_kernel<C>), dim3((unsigned)(nbm * nbn)), dim3(C::NT), C::LDS_BYTES, s, p);
  return check_launch();
}

// ================================================================================================================
// The same tile, ring and K-step order for BF6 ("F6") operands (ATOM_AB_F6 | ATOM_B_F6S: what the fused quantisers emit and what
// atom_repack_weight_f6s makes of a weight): the instruction count of a K step is what a gfx950 SIMD pays for (every issue slot adds:
// profiles/r02, r05/mid_ab.txt -- the packed-format kernel above spends ~120 instructions per wave and step, 36 of them widening
// nibbles, 13 converting integers and halves, and runs a step in ~1,000 cycles), and in this format a step is 2 MFMAs
// (v_mfma_f32_16x16x128_f8f6f4: a whole group of a 16 x 16 micro-tile each, exact), 12-16 FP32 instructions of de-quantisation, ten
// 8-byte fragment loads and two LDS-DMA instructions per wave.  Stage: the tile's 64 weight records and 64 token records of the group
// -- each 6,656 contiguous bytes of the group-major format, lane-linear LDS-DMA, no per-row addresses -- and its 64 float32 weight
// scales; token scales ride in the records (byte 100).  Fragment rows interleaved as in the 256x256 kernel (micro-tile row i of
// fragment (h, k) = record 32 h + 2 i + k: conflict-free ds_read_b64 at the 104-byte pitch, 8 consecutive features per lane).  The
// keeper's two halves are staged exactly as above (INT8 pieces in MFMA order).
namespace f6m {

typedef int v8i __attribute__((ext_vector_type(8)));
constexpr int PITCH = 104;
                                                  // (a tile's 64 records are 6,656 contiguous bytes: seven 1 KiB pieces, the last one half)
// TM / TN = token blocks of 16 / feature blocks of 32 per wave (the same 8 waves = 4 token-block slots x 2 feature halves): 1 x 1 =
// 64 x 64 tiles; 2 x 1 = 128 x 64 (two token blocks per wave, tq and tq + 4: 4 MFMAs per wave and step on 20 KB of LDS-DMA instead of 2 on
// 14.5 KB); 2 x 2 = 128 x 128 (+ feature halves h and h + 2: 8 MFMAs on 26.5 KB).  The step of this kernel is bound by what a CU's
// LDS-DMA path moves and by the issue of the LDS-DMA pieces, so a shape takes the largest tile that still gives
// (nearly) every CU one.  Same K order whatever the tile: the bits do not change.
template <int NS_, bool PAIR_, int ABL_ = 0, bool S16_ = false, int TM_ = 1, int TN_ = 1>
struct Cfg {
  static constexpr int NW = 8, NS = NS_, NT = 512, TM = TM_, TN = TN_, BMT = 64 * TM_, BNT = 64 * TN_;
  // a tile side of 64 records is 6,656 contiguous bytes = seven 1 KiB LDS-DMA pieces (the last one half); of 128: thirteen
  static constexpr int W_PIECES = TN_ == 1 ? 7 : 13, A_PIECES = TM_ == 1 ? 7 : 13;
  static constexpr int W_OFF = 0, A_OFF = W_PIECES * 1024, SB_OFF = A_OFF + A_PIECES * 1024;
  static constexpr int STAGE = SB_OFF + 256 * TN_;                       // + the tile's float32 weight scales
  // keeper halves in the same slot: the packed kernel's layout (4 TN weight fragments + 4 TM token blocks in MFMA order, fp16 weight
  // scales, token scales as dwords)
  static constexpr int K_W = 0, K_A = 4096 * TN_, K_SB = K_A + 4096 * TM_, K_SA = K_SB + 256;
  static_assert(K_SA + 256 * TM_ <= STAGE, "the keeper's pieces fit a stage");
  // ring slots whose byte offset fits the 16-bit instruction offset of a DS load behind ONE address register; further slots use a second
  // register set (+ SPB stages)
  static constexpr int SPB = (65535 - PITCH - 24) / STAGE + 1 < NS_ ? (65535 - PITCH - 24) / STAGE + 1 : NS_;
  static_assert(SPB >= 2 && 2 * SPB >= NS_, "at most two address-register sets");
  static constexpr bool PAIR = PAIR_;
  static constexpr bool S16 = S16_;                // weights without appended float32 scales (atom_repack_weight_f6): the dense fp16 array, converted per step
  static_assert(!(S16_ && TN_ > 1), "the 128-feature tile takes float32 weight scales only");
  static constexpr int ABL = ABL_;                 // tools build only: 1 no LDS-DMA in the K loop, 2 no barrier, 4 no MFMA, 8 no LDS loads, 16 no de-quantisation
  static constexpr int PPW = TM_ + TN_;            // DMA instructions per wave and stage: waves 0..6 the weight / token pieces w (, 7 + w), wave 7 the scales
  static constexpr int LDS_BYTES = NS * STAGE;
  static_assert(NS >= 3 && PPW * (NS - 2) < 64 && LDS_BYTES <= 160 * 1024, "ring depth");
};

// LDS reads of the K loop through inline asm: an address register + an immediate offset each, and no pairing of a fragment's three
// 8-byte loads into ds_read2_b64 (half the LDS rate; what hipcc makes of adjacent loads).  The compiler does not see these as memory
// operations: the loop waits for them itself (s_waitcnt lgkmcnt(0) ahead of the MFMAs) and the step's s_barrier orders them against
// the LDS-DMA.
template <int OFF>
__device__ __forceinline__ v2u lds64(unsigned a) {
  v2u r;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ float lds32f(unsigned a) {
  float r;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ v4f lds128f(unsigned a) {
  v4f r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF));
  return r;
}
template <bool PAIR_, bool KEEPER_>
struct DeqTag { static constexpr bool value = PAIR_, keeper = KEEPER_; };   // what a de-quantisation runs on: a BF6 step (shared products?) or the keeper

struct KeeperDma {         // per lane: the keeper pieces of the packed kernel (row * 128 + 16 * source chunk), the wave's operand and scales
  unsigned kvoff, svoff;
  const uint8_t *d8;
  const half_t *s8;
  unsigned kvoff_w, dst, dst_w, sdst;   // TM = 2: every wave a token block (kvoff -> dst) AND a weight fragment (kvoff_w -> dst_w); scale piece -> sdst
};

template <class C>
__global__ __launch_bounds__(C::NT) void gemm_w4a4_f6m_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NS = C::NS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < 8);
  // wave tile: token block(s) tb (, tb + 4) of 16 tokens x feature block(s) h (, h + 2) of 32 features
  const int h = wave & 1, tb = wave >> 1;
  constexpr int TM = C::TM, TN = C::TN, STAGE = C::STAGE, W_OFF = C::W_OFF, A_OFF = C::A_OFF, SB_OFF = C::SB_OFF;
  constexpr int K_W = C::K_W, K_A = C::K_A, K_SB = C::K_SB, K_SA = C::K_SA, PPW = C::PPW;
  const int nbm = (p.M + C::BMT - 1) / C::BMT, nbn = p.N / C::BNT;
  int id = blockIdx.x;
  {
    const int nwg = nbm * nbn;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m0 = (id % nbm) * C::BMT, n0 = (id / nbm) * C::BNT;
  const int G = p.G, T = G + 2;

  // ---- LDS-DMA, loop invariant.  int4 stage g: waves 0..6 piece w of the weight records and piece w of the token records (piece 6: 32
  // lanes when the side is 64 records; a side of 128 records has 13 full pieces: waves 0..5 also take piece 7 + w, wave 6 repeats its
  // piece), wave 7 the float32 weight scales (repeated: the count per wave stays uniform at PPW)
  const uint8_t *wsrc0 = p.B4 + (int64_t)n0 * PITCH, *asrc0 = p.A4 + (int64_t)m0 * PITCH;
  const int64_t wstep = p.f6_rows_b * PITCH, astep = p.f6_rows_a * PITCH;
  const float *sbsrc0 = C::S16 ? reinterpret_cast<const float *>(p.sB) : p.sB32 + n0;   // (S16: fp16 [G][N]; 128 bytes per tile and group)
  const int64_t sbstep = C::S16 ? p.N / 2 : p.f6_rows_b;                                   // floats between groups
  const unsigned sbvoff = C::S16 ? (unsigned)min(n0 + 2 * lane, p.N - 2) * 2u : (unsigned)lane * 4u;
  const unsigned voff = (unsigned)(wave * 1024 + lane * 16);
  const int wave2 = wave < 6 ? 7 + wave : wave;            // the wave's second piece of a 13-piece side
  const unsigned voff2 = (unsigned)(wave2 * 1024 + lane * 16);
  const bool half_ok = wave < 6 || (wave == 6 && lane < 32);   // piece w of a 7-piece side exists (piece 6: its first half)
  KeeperDma kd;
  {
    const int i = lane >> 2, j = lane & 3;
    const unsigned chunk = (unsigned)((j ^ ((i >> 1) & 3)) << 4);
    if constexpr (TM == 1) {
      const bool is_w = wave < 4;
      const unsigned row = is_w ? (unsigned)(n0 + 32 * (wave >> 1) + 2 * i + (wave & 1)) : (unsigned)min(m0 + 32 * ((wave - 4) >> 1) + ((wave - 4) & 1) + 2 * i, p.M - 1);
      kd.kvoff = row * (unsigned)kKeeper + chunk;          // (token blocks interleaved like the BF6 records: block b row i = token 32 (b >> 1) + (b & 1) + 2 i)
      kd.d8 = is_w ? p.B8 : p.A8;
      kd.kvoff_w = 0; kd.dst = (unsigned)wave * 1024u; kd.dst_w = 0; kd.sdst = (wave & 1) ? K_SA : K_SB;
    } else {                                               // wave w: token block w (8 of them) and weight fragment w (TN = 1: w & 3 -- waves 4..7 repeat 0..3: same bytes, same place)
      const int f = TN == 2 ? wave : (wave & 3);
      kd.kvoff = (unsigned)min(m0 + 32 * (wave >> 1) + (wave & 1) + 2 * i, p.M - 1) * (unsigned)kKeeper + chunk;
      kd.kvoff_w = (unsigned)(n0 + 32 * (f >> 1) + 2 * i + (f & 1)) * (unsigned)kKeeper + chunk;
      kd.d8 = p.A8;
      kd.dst = K_A + (unsigned)wave * 1024u; kd.dst_w = K_W + (unsigned)f * 1024u;
      kd.sdst = (wave & 1) ? K_SA + ((wave >> 1) & 1) * 256u : K_SB;      // waves 1, 5: tokens 0..63; waves 3, 7: tokens 64..127
    }
    if (wave & 1) {
      const int m = min(m0 + (TM == 2 ? ((wave >> 1) & 1) * 64 : 0) + lane, p.M - 1);
      kd.svoff = (unsigned)(p.ref_layout ? ref_scale_index(m) : m) * 2u;
      kd.s8 = p.sA8;
    } else {
      kd.svoff = (unsigned)min(n0 + 2 * lane, p.N - 2) * 2u;     // (one dword piece = 128 halves: all the weight scales of a 128-feature tile)
      kd.s8 = p.sB8;
    }
  }
  const unsigned lds0 = lds_addr(lds);
  auto issue_int4 = [&](const uint8_t *wsrc, const uint8_t *asrc, const float *sbsrc, unsigned slot) {
    if (wave < 7) {
      if constexpr (TN == 1) {
        if (half_ok) lds_dma_sv<16>(wsrc, voff, slot + W_OFF + wave * 1024);
      } else {
        lds_dma_sv<16>(wsrc, voff, slot + W_OFF + wave * 1024);
        lds_dma_sv<16>(wsrc, voff2, slot + W_OFF + wave2 * 1024);
      }
      if constexpr (TM == 1) {
        if (half_ok) lds_dma_sv<16>(asrc, voff, slot + A_OFF + wave * 1024);
      } else {
        lds_dma_sv<16>(asrc, voff, slot + A_OFF + wave * 1024);
        lds_dma_sv<16>(asrc, voff2, slot + A_OFF + wave2 * 1024);
      }
    } else {
#pragma unroll
      for (int i = 0; i < PPW; ++i)                        // (TN = 2: 128 floats = two dword pieces, then repeats)
        lds_dma_sv<4>(sbsrc + (TN == 2 ? 64 * (i & 1) : 0), sbvoff, slot + SB_OFF + (TN == 2 ? 256 * (i & 1) : 0));
    }
  };
  auto issue = [&](int st, unsigned slot) {                // any stage (prologue and the last NS - 1 steps)
    if (st < G) {
      issue_int4(wsrc0 + st * wstep, asrc0 + st * astep, sbsrc0 + (int64_t)st * sbstep, slot);
    } else {                                               // keeper half st - G: the wave's piece(s) of the packed kernel's stage + a scale piece
      lds_dma_sv<16>(kd.d8 + (st - G) * 64, kd.kvoff, slot + kd.dst);
      if constexpr (TM == 2) lds_dma_sv<16>(p.B8 + (st - G) * 64, kd.kvoff_w, slot + kd.dst_w);
#pragma unroll
      for (int i = 0; i < PPW - TM; ++i) {                 // (the count per wave stays PPW)
        if (wave & 1) lds_dma_sv<2>(kd.s8, kd.svoff, slot + kd.sdst);
        else lds_dma_sv<4>(kd.s8, kd.svoff, slot + kd.sdst);
      }
    }
  };
#pragma unroll 1
  for (int s = 0; s < min(NS - 1, T); ++s) issue(s, lds0 + s * STAGE);

  const int l15 = lane & 15, kb = lane >> 4;
  // fragment (hh, k) row l15 = record 32 hh + 2 l15 + k; token block b row l15 = record 32 (b >> 1) + (b & 1) + 2 l15; a wave's second
  // feature block (h + 2) and second token block (tb + 4) are the same rows 64 records further
  constexpr int R64 = 64 * PITCH;                          // byte distance of 64 records
  constexpr int SPB = C::SPB, NB = (NS + SPB - 1) / SPB;   // ring slots per address-register set, sets
  unsigned aw[NB], aa[NB], as_[NB], asb[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const unsigned base = lds0 + b * SPB * STAGE;
    aw[b] = base + W_OFF + (32 * h + 2 * l15) * PITCH + kb * 24;                            // + PITCH: k = 1
    aa[b] = base + A_OFF + (32 * (tb >> 1) + (tb & 1) + 2 * l15) * PITCH + kb * 24;
    as_[b] = base + A_OFF + (32 * (tb >> 1) + (tb & 1) + 2 * l15) * PITCH + 100;            // the token's float32 scale
    asb[b] = base + SB_OFF + (32 * h + 8 * kb) * (C::S16 ? 2 : 4);
  }

  float c[TM][TN][2][4];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[t][n][k][r] = 0.f;
  // what a step leaves for the next one to de-quantise, in two register sets used alternately (the loop below is unrolled by two: no
  // copies, and a step's scales are requested ahead of the previous step's de-quantisation instead of behind it)
  struct Pend { v4f acc[TM][TN][2]; float sa[TM]; v4f sb[TN][2]; };
  Pend P[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int n = 0; n < TN; ++n) P[i].sb[n][0] = P[i].sb[n][1] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TM; ++t) {
#pragma unroll
      for (int n = 0; n < TN; ++n) P[i].acc[t][n][0] = P[i].acc[t][n][1] = v4f{0.f, 0.f, 0.f, 0.f};
      P[i].sa[t] = 0.f;
    }
  }
  // c[t][n][k][r] = fma(idot, sa * sb, c): token = the lane's of block t, feature 32 (h + 2 n) + 8 kb + 2 r + k (the contract; PAIR: one
  // product per channel pair)
  auto dequant = [&](const Pend &q, auto pair) {
    constexpr bool PR = decltype(pair)::value;
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      float sbv[8];
      if constexpr (C::S16 && !decltype(pair)::keeper) {   // eight halves in sb[0][0]
        const half_t *hv = reinterpret_cast<const half_t *>(&q.sb[n][0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) sbv[j] = (float)hv[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sbv[j] = q.sb[n][0][j]; sbv[4 + j] = q.sb[n][1][j]; }
      }
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        float s[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[2 * r] = q.sa[t] * sbv[2 * r];
          s[2 * r + 1] = PR ? s[2 * r] : q.sa[t] * sbv[2 * r + 1];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            c[t][n][k][r] = __builtin_fmaf(q.acc[t][n][k][r], s[2 * r + k], c[t][n][k][r]);
            asm volatile("" : "+v"(c[t][n][k][r]));
          }
      }
    }
  };
  // One K step on the stage in ring slot SL (compile time: the slot's byte offset rides in the loads' immediate offsets -- no address
  // arithmetic at all) or at the run-time byte offset `so` (SL < 0: the last steps).  Every LDS load of the step is requested first
  // (fragments, token scale, weight scales -- into the register set `cur`), the LDS-DMA of the stage NS - 1 ahead is issued behind
  // them (its ~100 issue cycles per piece overlap the loads' latency), the previous step's products (`prv`) are de-quantised, then
  // this step's MFMAs (consumed in the next step).
  auto compute = [&](auto slc, unsigned so, Pend &cur, const Pend &prv, auto dma) {
    constexpr int SL = decltype(slc)::value;
    constexpr int SET = SL < 0 ? 0 : SL / SPB;
    constexpr int O = SL < 0 ? 0 : (SL % SPB) * STAGE;
    const unsigned xw = SL < 0 ? aw[0] + so : aw[SET], xa = SL < 0 ? aa[0] + so : aa[SET];
    const unsigned xs = SL < 0 ? as_[0] + so : as_[SET], xb = SL < 0 ? asb[0] + so : asb[SET];
    v2u fw[TN][6], fa[TM][3];                              // weight fragments k = 0, 1 per feature block, a token fragment per block: three 8-byte pieces each
    if constexpr (!(C::ABL & 8)) {
      fw[0][0] = lds64<O>(xw); fw[0][1] = lds64<O + 8>(xw); fw[0][2] = lds64<O + 16>(xw);
      fw[0][3] = lds64<O + PITCH>(xw); fw[0][4] = lds64<O + PITCH + 8>(xw); fw[0][5] = lds64<O + PITCH + 16>(xw);
      fa[0][0] = lds64<O>(xa); fa[0][1] = lds64<O + 8>(xa); fa[0][2] = lds64<O + 16>(xa);
      cur.sa[0] = lds32f<O>(xs);
      if constexpr (TM == 2) {                             // (their own address registers: a slot offset + 6,656 bytes can exceed 16 bits)
        const unsigned xa2 = xa + R64, xs2 = xs + R64;
        fa[TM - 1][0] = lds64<O>(xa2); fa[TM - 1][1] = lds64<O + 8>(xa2); fa[TM - 1][2] = lds64<O + 16>(xa2);
        cur.sa[TM - 1] = lds32f<O>(xs2);
      }
      if constexpr (TN == 2) {
        const unsigned xw2 = xw + R64;
        fw[TN - 1][0] = lds64<O>(xw2); fw[TN - 1][1] = lds64<O + 8>(xw2); fw[TN - 1][2] = lds64<O + 16>(xw2);
        fw[TN - 1][3] = lds64<O + PITCH>(xw2); fw[TN - 1][4] = lds64<O + PITCH + 8>(xw2); fw[TN - 1][5] = lds64<O + PITCH + 16>(xw2);
      }
      cur.sb[0][0] = lds128f<O>(xb);
      if constexpr (!C::S16) cur.sb[0][1] = lds128f<O + 16>(xb);
      if constexpr (TN == 2) {
        cur.sb[TN - 1][0] = lds128f<O + 256>(xb);
        cur.sb[TN - 1][1] = lds128f<O + 256 + 16>(xb);
      }
    } else {
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int i = 0; i < 6; ++i) fw[n][i] = v2u{so + i, so};
#pragma unroll
      for (int t = 0; t < TM; ++t) {
#pragma unroll
        for (int i = 0; i < 3; ++i) fa[t][i] = v2u{so + i, so + 7};
        cur.sa[t] = 1.f;
      }
#pragma unroll
      for (int n = 0; n < TN; ++n) cur.sb[n][0] = cur.sb[n][1] = v4f{1.f, 1.f, 1.f, 1.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(C::ABL & 1)) dma();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(C::ABL & 16)) dequant(prv, DeqTag<C::PAIR, false>());
    else { c[0][0][0][0] += prv.acc[0][0][0][0] + prv.acc[TM - 1][TN - 1][1][3]; asm volatile("" : "+v"(c[0][0][0][0])); }
    // The loads above are invisible to the compiler: until they have landed, their destination registers must stay allocated and
    // unread -- also the components nothing reads later (PAIR uses every other weight scale), which the register allocator would
    // otherwise hand to the de-quantisation's temporaries while the load is still in flight.  Every destination is an in/out operand
    // of the wait itself (or of the empty statements behind it, which the scheduling barrier keeps behind it): nothing is copied or
    // re-used before it.
#define ATOM_F6M_BASE "+v"(fw[0][0]), "+v"(fw[0][1]), "+v"(fw[0][2]), "+v"(fw[0][3]), "+v"(fw[0][4]), "+v"(fw[0][5]), "+v"(fa[0][0]), "+v"(fa[0][1]), \
                      "+v"(fa[0][2]), "+v"(cur.sa[0]), "+v"(cur.sb[0][0]), "+v"(cur.sb[0][1])
#define ATOM_F6M_T2 "+v"(fa[TM - 1][0]), "+v"(fa[TM - 1][1]), "+v"(fa[TM - 1][2]), "+v"(cur.sa[TM - 1])
#define ATOM_F6M_N2 "+v"(fw[TN - 1][0]), "+v"(fw[TN - 1][1]), "+v"(fw[TN - 1][2]), "+v"(fw[TN - 1][3]), "+v"(fw[TN - 1][4]), "+v"(fw[TN - 1][5]), \
                    "+v"(cur.sb[TN - 1][0]), "+v"(cur.sb[TN - 1][1])
    if constexpr (TM == 1 && TN == 1) asm volatile("s_waitcnt lgkmcnt(0)" : ATOM_F6M_BASE::"memory");
    else if constexpr (TN == 1) asm volatile("s_waitcnt lgkmcnt(0)" : ATOM_F6M_BASE, ATOM_F6M_T2::"memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : ATOM_F6M_BASE, ATOM_F6M_T2, ATOM_F6M_N2::"memory");
#undef ATOM_F6M_BASE
#undef ATOM_F6M_T2
#undef ATOM_F6M_N2
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const v8i w0 = {(int)fw[n][0].x, (int)fw[n][0].y, (int)fw[n][1].x, (int)fw[n][1].y, (int)fw[n][2].x, (int)fw[n][2].y, 0, 0};
      const v8i w1 = {(int)fw[n][3].x, (int)fw[n][3].y, (int)fw[n][4].x, (int)fw[n][4].y, (int)fw[n][5].x, (int)fw[n][5].y, 0, 0};
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const v8i bf = {(int)fa[t][0].x, (int)fa[t][0].y, (int)fa[t][1].x, (int)fa[t][1].y, (int)fa[t][2].x, (int)fa[t][2].y, 0, 0};
        if constexpr (!(C::ABL & 4)) {
          cur.acc[t][n][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w0, bf, v4f{0.f, 0.f, 0.f, 0.f}, 3, 3, 0, 0, 0, 0);
          cur.acc[t][n][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w1, bf, v4f{0.f, 0.f, 0.f, 0.f}, 3, 3, 0, 0, 0, 0);
        } else {
          cur.acc[t][n][0] = v4f{(float)w0[0], (float)w0[5], (float)bf[1], (float)w1[2]};
          cur.acc[t][n][1] = v4f{(float)w1[0], (float)w1[5], (float)bf[3], (float)bf[5]};
        }
      }
    }
  };

  // `allowed` stages of this wave's LDS-DMA may still be in flight (everything older has landed), then the workgroup barrier: every
  // wave's pieces of the step's stage are in LDS, and every wave is done reading the slot of the previous step
  auto sync = [&](int allowed) {
    if constexpr (!(C::ABL & 1)) {
      if (allowed >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NS - 2)) : "memory");
      else if (allowed == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      else if (allowed == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * 2) : "memory");
      else if (allowed == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * 3) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (!(C::ABL & 2)) __builtin_amdgcn_s_barrier();
  };
  static_assert(NS - 2 <= 4, "sync(): one immediate per possible count");
  int step = 0;
  if constexpr (NS == 4) {
    // steps whose LDS-DMA is an int4 stage, NS per trip: ring slots, register sets and DMA destinations are compile-time constants,
    // the operand pointers run
    const uint8_t *wsrc = wsrc0 + (NS - 1) * wstep, *asrc = asrc0 + (NS - 1) * astep;
    const float *sbsrc = sbsrc0 + (int64_t)(NS - 1) * sbstep;
    auto one = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      sync(NS - 2);
      const unsigned dst = lds0 + ((k + NS - 1) % NS) * STAGE;
      auto dma = [&]() {
        issue_int4(wsrc, asrc, sbsrc, dst);
        wsrc += wstep; asrc += astep; sbsrc += sbstep;
      };
      compute(std::integral_constant<int, k>(), 0u, P[k & 1], P[(k & 1) ^ 1], dma);
    };
    static_assert(NS == 4, "the unrolled trip is written for a ring of four");
    while (step + 2 * NS - 2 < G) {                        // the trip's last request is stage step + 2 NS - 2
      one(std::integral_constant<int, 0>());
      one(std::integral_constant<int, 1>());
      one(std::integral_constant<int, 2>());
      one(std::integral_constant<int, 3>());
      step += NS;
    }
  }
  // the remaining steps (their LDS-DMA: the last int4 stages, then the keeper's halves, then nothing: the waits count what is in flight)
  while (step < G) {
    const int slot = step % NS;
    sync(min(NS - 2, T - step - 1));
    auto dma = [&]() {
      if (step + NS - 1 < T) issue(step + NS - 1, lds0 + ((step + NS - 1) % NS) * STAGE);
    };
    if (step & 1) compute(std::integral_constant<int, -1>(), (unsigned)(slot * STAGE), P[1], P[0], dma);
    else compute(std::integral_constant<int, -1>(), (unsigned)(slot * STAGE), P[0], P[1], dma);
    ++step;
  }
  if (G & 1) dequant(P[0], DeqTag<C::PAIR, false>());
  else dequant(P[1], DeqTag<C::PAIR, false>());
  const int slot = G % NS;
  // the keeper (INT8, both halves in one step, per-channel scales)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    const char *s0 = lds + slot * STAGE, *s1 = lds + (slot + 1 == NS ? 0 : slot + 1) * STAGE;
    const int coff = l15 * 64 + ((kb ^ ((l15 >> 1) & 3)) << 4);
    Pend kq;
    v4i b0[TM], b1[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int b = tb + 4 * t;                            // the token block
      const int ka = K_A + b * 1024 + coff;
      b0[t] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(s0 + ka));
      b1[t] = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(s1 + ka));
      // token scale: the lane's token is record row 32 (b >> 1) + (b & 1) + 2 l15 of the tile = index of its dword in the scale pieces
      const unsigned sah = *reinterpret_cast<const unsigned *>(s0 + K_SA + (32 * (b >> 1) + (b & 1) + 2 * l15) * 4);
      kq.sa[t] = (float)__builtin_bit_cast(half_t, (unsigned short)sah);
    }
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int hh = h + 2 * n;                            // the feature block: fragments 2 hh + k
      const int kw = K_W + 2 * hh * 1024 + coff;
      const v4u sbv = *reinterpret_cast<const v4u *>(s0 + K_SB + (32 * hh + 8 * kb) * 2);
      const half_t *hv = reinterpret_cast<const half_t *>(&sbv);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const v4i x0 = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(s0 + kw + k * 1024));
        const v4i x1 = __builtin_bit_cast(v4i, *reinterpret_cast<const v4u *>(s1 + kw + k * 1024));
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          v4i a = __builtin_amdgcn_mfma_i32_16x16x64_i8(x0, b0[t], v4i{0, 0, 0, 0}, 0, 0, 0);
          a = __builtin_amdgcn_mfma_i32_16x16x64_i8(x1, b1[t], a, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) kq.acc[t][n][k][r] = (float)a[r];
        }
      }
      kq.sb[n][0] = v4f{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
      kq.sb[n][1] = v4f{(float)hv[4], (float)hv[5], (float)hv[6], (float)hv[7]};
    }
    dequant(kq, DeqTag<false, true>());
  }
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int b = tb + 4 * t;
    const int m = m0 + 32 * (b >> 1) + (b & 1) + 2 * l15;
    if (m < p.M) {
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        v4u o;
        half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ov[2 * r] = f2h(c[t][n][0][r]);
          ov[2 * r + 1] = f2h(c[t][n][1][r]);
        }
        *reinterpret_cast<v4u *>(p.D + (int64_t)m * p.N + n0 + 32 * (h + 2 * n) + 8 * kb) = o;
      }
    }
  }
}

template <class C>
static int launch(const GemmParams &p, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  if (ensure_max_lds(reinterpret_cast<const void *>(&gemm_w4a4_f6m_kernel<C>), C::LDS_BYTES, attr_done) != ATOM_OK) return ATOM_ERR_LAUNCH;
  const int nbm = (p.M + C::BMT - 1) / C::BMT, nbn = p.N / C::BNT;
  hipLaunchKernelGGL((gemm_w4a4_f6m_kernel<C>), dim3((unsigned)(nbm * nbn)), dim3(C::NT), C::LDS_BYTES, s, p);
  return check_launch();
}

}  // namespace f6m
}  // namespace mid

// The tile of the BF6 mid-size-batch kernel for a shape: 1 = 64 x 64 (up to 256 of them), 2 = 128 x 64 (257 .. 512 tiles of 64 x 64:
// otherwise two workgroups per CU).  3 = 128 x 128 exists in the tuning build only (ATOM_MID_TM=3; float32 weight scales shared by
// channel pairs, N % 128 == 0): on shapes with at most 256 such tiles it measures 2-7 % ahead of the two-wave-group 128 x 128 kernel
// that serves them (768 x 4096 x 4096 20.8 -> 19.6 us, 1024 x .. 22.4 -> 22.0, 256 x 13824 x 5120 25.4 -> 24.5) and behind everywhere
// else (profiles/r05/mid_tm3.txt) -- not worth a summation order that would depend on the weight's flavour.
int f6_mid_tile(int64_t M, int64_t N, bool f32_scales) {
  const int64_t t64 = ((M + 63) / 64) * (N / 64);
  int t = t64 <= 256 ? 1 : 2;
#ifdef ATOM_TOOLS
  if (const int forced = ATOM_TUNE("ATOM_MID_TM", 0)) t = forced == 3 && (!f32_scales || (N % 128) != 0) ? 2 : forced;
#endif
  return t;
}

// BF6 operands (ATOM_AB_F6; with the appended float32 weight scales of ATOM_B_F6S or the dense fp16 array), fp16 output, N % 64 == 0
int launch_gemm_f6_mid(const GemmParams &p, hipStream_t s) {
  if (!p.f6_rows_a || (p.N % mid::BN) != 0 || !p.D) return ATOM_ERR_SHAPE;
  using namespace mid::f6m;
  // the tile: 64 x 64 up to 256 of them; 128 x 64 for 257 .. 512 tiles of 64 x 64 (otherwise two workgroups per CU); 128 x 128 beyond
  // (f6_pick_cfg sends such shapes here only where that leaves at most one tile per CU)
  const int tsel = f6_mid_tile(p.M, p.N, p.sB32 != nullptr);
  if (!p.sB32) return tsel == 2 ? launch<Cfg<4, false, 0, true, 2>>(p, s) : launch<Cfg<4, false, 0, true>>(p, s);
#ifdef ATOM_TOOLS
  if (tsel == 3 && p.b_pairs) return launch<Cfg<4, true, 0, false, 2, 2>>(p, s);   // (without shared scale products the 128 x 128 wave tile spills: 128 x 64 then)
#endif
  if (tsel >= 2) return p.b_pairs ? launch<Cfg<4, true, 0, false, 2>>(p, s) : launch<Cfg<4, false, 0, false, 2>>(p, s);
#ifdef ATOM_TOOLS
  switch (ATOM_TUNE("ATOM_MID_ABL", 0)) {
#define ATOM_ABL(a) case a: return launch<Cfg<4, true, a>>(p, s);
    ATOM_ABL(1) ATOM_ABL(2) ATOM_ABL(3) ATOM_ABL(4) ATOM_ABL(8) ATOM_ABL(16) ATOM_ABL(27) ATOM_ABL(31)
#undef ATOM_ABL
    default: break;
  }
#endif
  return p.b_pairs ? launch<Cfg<4, true>>(p, s) : launch<Cfg<4, false>>(p, s);
}

namespace mid {
}  // namespace mid

// Packed operands (reference format), fp16 output, N % 64 == 0 (the library's constraint anyway).  8 waves (16 tokens x 32 features
// each: two waves per SIMD keep its VALU fed) on an 8-stage ring (profiles/r05/mid_ab.txt: 16 stages measure 1 us slower -- the
// prologue issues them all --, 4 waves of 32 x 32 a third slower at one tile per CU).
int launch_gemm_mid(const GemmParams &p, hipStream_t s) {
  if (p.a_wide || p.f6_rows_a || (p.N % mid::BN) != 0 || !p.D) return ATOM_ERR_SHAPE;
#ifdef ATOM_TOOLS
  const int nw = ATOM_TUNE("ATOM_MID_NW", 8), ns = ATOM_TUNE("ATOM_MID_NS", 8);
  if (nw == 8 && ns == 16) return p.b_pairs ? mid::launch<mid::Cfg<8, 16, true>>(p, s) : mid::launch<mid::Cfg<8, 16, false>>(p, s);
  if (nw == 4 && ns == 8) return p.b_pairs ? mid::launch<mid::Cfg<4, 8, true>>(p, s) : mid::launch<mid::Cfg<4, 8, false>>(p, s);
#endif
  return p.b_pairs ? mid::launch<mid::Cfg<8, 8, true>>(p, s) : mid::launch<mid::Cfg<8, 8, false>>(p, s);
}

}  // namespace atom
