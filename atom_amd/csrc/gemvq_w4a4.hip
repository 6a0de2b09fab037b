// W4A4 GEMM, one or two tokens of a decode step, with the quantiser that feeds it INSIDE the launch (round 6).
//
// A decode layer at batch 1 is four (quantiser -> projection) pairs around the attention (reference call order
// e2e/punica-atom/punica/models/llama.py:259-292, :85-87): RMSNorm -> q / k / v, reorder -> o_proj, residual add + RMSNorm -> gate / up,
// SiLU x up -> down_proj.  As separate launches each quantiser costs 4.4-4.5 us of a 60 us layer for ONE row of work (a chain of cold
// dependent loads on a single workgroup, then a launch boundary); inside the decode-batch kernel (gemm_w4a4_skinny.hip, rounds 3-5) every
// one of its N / 16 workgroups repeated the quantiser and its one-block-deep weight prefetch ran the large projections at 1.8 TB/s
// (profiles/r06/decode_prof_*.txt: gate / up 24.8 us fused against 4.5 + 11.1 separate).  This kernel is the dot-product kernel
// (gemv_w4a4.hip gemv1_w4a4_kernel: one wave per output feature, lanes along K, v_dot8_i32_i4 on the packed dwords) restructured so
// that the quantiser runs ONCE PER CU:
//   * grid = one workgroup of 16 waves per CU (at most); a workgroup owns a contiguous range of the output features (XCD-contiguous:
//     a 64-byte line of weight scales serves 32 adjacent features), wave w of it the features w, w + 16, ... of that range;
//   * the quantiser's own requests (token rows, norm weights, reorder indices) go out FIRST, in one batch, and one s_barrier keeps
//     every weight request behind them: the chip hands a CU about 10 bytes per clock, so whatever is requested first is what arrives
//     first (with the weight ring in front, the quantiser's inputs reached LDS after 6-12 k cycles: profiles/r06/gemvq_trace.txt);
//   * at one token the waves have roles (q_roles): waves 0-7 run the quantiser and synchronise through an LDS counter (qsync), waves
//     8-15 request their first D feature steps right behind the barrier and spin on that counter until the packed operand -- INT4
//     codes, INT8 keeper, fp16 scales: the bytes the stand-alone quantiser kernels write -- is in LDS; the quantiser waves request their
//     features behind their last qsync (the split merge, q_op 5, too, when its row fits the eight quantiser waves).  At two tokens, and
//     for SiLU x up, whose row needs all 1024 threads, every wave does both jobs with s_barrier in qsync's place
//     (profiles/r06/ab_gemvq_roles.txt, ab_gemvq_roles4.txt);
//   * the feature loop keeps a ring of D steps in flight per wave (a step = a whole, a half or a quarter of one feature's chunks:
//     PartW below), refilled in place inside a loop unrolled by D, reads the token's codes out of LDS (one ds_read_b128 per weight
//     chunk) and forms every sum exactly as gemv1_w4a4_kernel does: lane l owns chunks l, l + 64, ... in ascending order, a quad sums a
//     group exactly, the quad leader applies c = fma(idot, sA * sB, c), 64-lane butterfly, keeper last.
// Output: bit-identical to the stand-alone quantiser launch followed by atom_gemm_w4a4_multi (which runs gemv1_w4a4_kernel for these
// token counts) -- atom_gemm_w4a4_packed_order() = 64 on both sides.
// Quantiser arithmetic: the kernel-flavoured mode of quant_kernels.hip slot by slot (Reorder.cuh:137-178, RMSNorm.cuh:112-151,
// Activate.cuh:112-167), the sum of squares as the same fixed-shape FP32 tree (256 threads per row).
#include <type_traits>
#include "common.h"
#include "quant_math.h"

namespace atom {
namespace gemvq {

constexpr int NTH = 1024, NWV = NTH / 64;
constexpr int MQ = 2;                     // token rows at most
constexpr int CPT = 8;                    // channels per quantiser task: a 128-channel group = 16 adjacent lanes (one DPP row)
constexpr int TPT1 = 2, XC = 2;           // quantiser tasks per thread and token row / 16-byte row chunks per thread at most (gemvq_fits)

__device__ __forceinline__ int quad_sum(int d) {
  d += __builtin_amdgcn_mov_dpp(d, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  d += __builtin_amdgcn_mov_dpp(d, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  return d;
}

// max over the aligned 16 lanes this lane belongs to (a quantisation group of 8-channel tasks = one DPP row): exact, order-free
__device__ __forceinline__ float max16(float a) {
  a = max8(a);
  a = fmaxf(a, dpp_f<0x140>(a));                      // row_mirror: lanes 8-15 <-> 7-0 of the row
  return a;
}

// LDS: the packed operand [MQ][K4h] codes, [MQ][128] keeper, [G][MQ] + [MQ] fp16 scales; then (16-byte aligned) the reduction scratch,
// and for ops 1-3 the fp16 rows [MQ][H] and the norm weights [H]
__host__ __device__ inline int red_offset(int K4h, int G) { return (MQ * K4h + MQ * kKeeper + MQ * G * 2 + MQ * 2 + 15) & ~15; }
constexpr int RED_BYTES = 48;             // [MQ][4] partial sums of squares, then the staging counter (+ padding)
__host__ __device__ inline int quant_lds_bytes(int q_op, int K4h, int G) {   // everything the quantiser uses
  const int H = 2 * K4h + kKeeper;
  return (red_offset(K4h, G) + RED_BYTES + (q_op != 4 ? H * 2 * (MQ + 1) : 0) + 15) & ~15;
}
inline size_t lds_bytes(int q_op, int K4h, int G) { return (size_t)quant_lds_bytes(q_op, K4h, G); }

// One step of a wave's feature loop: PCH chunks of ONE output feature (a whole feature up to 2 chunks per lane, a half or a quarter of
// one beyond: with 16 waves per workgroup a wave has 128 registers) and the scales of those chunks' groups.  The ring below holds D steps.
template <int PCH>
struct PartW {
  v4i w[PCH];              // chunks lane + 64 (part PCH + k), clamped to the row's last one
  unsigned short sbu[PCH]; // the weight scale of each chunk's group
};

// (`w8`, `sb8u`: the keeper chunk lane % 8 of the feature and its scale, requested with the feature's last part -- `last` is a constant
// once the loops are unrolled)
// (`addu`: with ADD, the fp16 addend of the feature's output -- the residual stream behind o_proj / down_proj -- requested with the
// weights instead of behind the last sum, where it was a trip to memory at the end of every wave; unconditional: a valid dummy address
// when the launch has no addend or the feature lies in another segment, so that the wait counts stay exact)
template <int PCH, bool ADD = false>
__device__ __forceinline__ void load_part(const GemmParams &p, int n, int part, bool last, int lane, int nchunks, PartW<PCH> &f, v4i &w8,
                                          unsigned short &sb8u, unsigned short *addu = nullptr) {
  const uint8_t *brow = p.B4 + (int64_t)n * p.K4h;
  const unsigned short *sBu = reinterpret_cast<const unsigned short *>(p.sB);
#pragma unroll
  for (int c = 0; c < PCH; ++c) {
    const int cc = min(lane + 64 * (part * PCH + c), nchunks - 1);
    f.w[c] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(brow + cc * 16));   // nt: read once, by this CU only
  }
#pragma unroll
  for (int c = 0; c < PCH; ++c) {
    const int cc = min(lane + 64 * (part * PCH + c), nchunks - 1);
    f.sbu[c] = sBu[(int64_t)(cc >> 2) * p.N + n];
  }
  if (last) {
    w8 = *reinterpret_cast<const v4i *>(p.B8 + (int64_t)n * kKeeper + (lane & 7) * 16);
    sb8u = reinterpret_cast<const unsigned short *>(p.sB8)[n];
    if constexpr (ADD) {
      const unsigned short *ap = (p.seg_add && n < p.seg_n) ? reinterpret_cast<const unsigned short *>(p.seg_add) + n
                                                            : reinterpret_cast<const unsigned short *>(p.sB8) + n;
      addu[0] = *ap;
    }
  }
}

// QOP: 1 reorder, 2 RMSNorm + reorder, 3 residual add + RMSNorm + reorder, 4 SiLU(x) * x2.  NCH: 16-byte weight chunks per lane
// (>= ceil(K4 / 2048)).  MT: token rows.
// OWN (with roles; launch1 picks it where a workgroup's whole share is at most two feature steps of its eight streamer waves -- N <= 4096:
// o_proj): the streamer waves own EVERY feature, the quantiser waves none -- everything is requested while the quantiser runs and
// nothing behind the operand's publication, where the quantiser waves' own single step was the tail of the launch (o_proj: done at
// 7.8 k instead of 8.5 k cycles).  With three steps per wave and more the eight streamers' arithmetic costs more than it saves
// (q / k / v: profiles/r06/ab_gemvq_own.txt).
template <int QOP, int NCH, int MT, int D, bool OWN = false>
__global__ __launch_bounds__(NTH) void gemvq_w4a4_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K4h = p.K4h, G = p.G;
  const int H = 2 * K4h + kKeeper;
  const int nchunks = K4h >> 4;
  // workgroup b runs on XCD b % 8: XCD-contiguous order, then a contiguous share of the features
  const int xq = (int)gridDim.x >> 3, xr = (int)gridDim.x & 7, xx = blockIdx.x & 7;
  const int lw = xx * xq + min(xx, xr) + ((int)blockIdx.x >> 3);
#ifdef ATOM_TOOLS   // tools/r06/gemvq_trace.py: s_memtime stamps of the first and the last workgroup into p.Dsz as u32 [2][16 waves][16]
  unsigned *trb = nullptr;
  if (p.Dsz && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) trb = reinterpret_cast<unsigned *>(p.Dsz) + ((blockIdx.x ? 16 : 0) + wave) * 16;
#define GQ_STAMP(k) do { if (trb) { __builtin_amdgcn_sched_barrier(0); const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); if (lane == 0) trb[k] = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
  if (trb && lane == 0) trb[15] = (unsigned)__builtin_amdgcn_s_memrealtime();
#else
#define GQ_STAMP(k) do { } while (0)
#endif
  GQ_STAMP(0);
  {  // the kernel arguments the prologue needs, in ONE batch of scalar loads (left to itself hipcc fetches each where it is first used: a
     // scalar-cache round trip in front of every group of requests)
    const void *a0 = p.q_x, *a1 = p.q_x2, *a2 = p.q_res, *a3 = p.q_idx, *a4 = p.B4, *a5 = p.sB, *a6 = p.q_res_out;
    const int i0 = p.M, i1 = p.K4h, i2 = p.G, i3 = p.N, i4 = p.q_roles, i5 = (int)gridDim.x;     // (gridDim: an implicit argument behind the struct)
    asm volatile("" ::"s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(a6), "s"(i0), "s"(i1), "s"(i2), "s"(i3), "s"(i4), "s"(i5));
  }
  // QOP 5 (round 6): the token rows are the decode attention's output still in KV-split form -- FP32 partial states [M][heads][splits]
  // [128 values, m, d] (csrc/kv_i4.hip) -- merged here exactly as decode_merge_kernel merges them, then reordered and quantised as QOP 1
  constexpr bool ROWS = QOP <= 3 || QOP == 5;               // ops that stage fp16 token rows in LDS (gathered through the reorder index)
  constexpr bool NORM = QOP == 2 || QOP == 3;
  constexpr int PS = 16, PH = 8;                            // QOP 5: KV splits at most (gemvq_merge_fits); their values in batches of PH

  // ---- Roles (round 6, p.q_roles: one token, ops 1-3): waves 0 .. NP - 1 are the workgroup's QUANTISER -- they stage the rows, run the sum
  // of squares and write the codes, synchronising among themselves through an LDS counter --, waves NP .. 15 are STREAMERS: they request
  // their first four feature steps at once and wait for the counter to say "operand published".  A CU takes ~10 bytes per clock from HBM
  // and a wave with 8 KB to request sits in issue for thousands of cycles (profiles/r06/gemvq_trace.txt): with every wave doing both, the
  // quantiser's barriers waited for waves stuck in issue (operand published at 12-16 k cycles); with s_barrier out of the way the
  // stream starts at ~1 k cycles on waves that have nothing else to do.  The quantiser waves request their own features behind the
  // last counter.  Same arithmetic on the same bytes either way.
  constexpr int NP = 8;
  const bool roles = (QOP <= 3 || QOP == 5) && MT == 1 && p.q_roles != 0;   // (workgroup-uniform)
  const bool streamer = roles && wave >= NP;
  const int PT = roles ? NP * 64 : NTH;                       // threads that share the quantiser's work
  unsigned *sync_cnt = reinterpret_cast<unsigned *>(lds + red_offset(K4h, G) + 32);
  int sync_no = 0;
  auto qsync = [&]() {                                        // barrier of the quantiser's waves: an LDS counter (roles) or s_barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (roles) {
      ++sync_no;
      if (lane == 0) __hip_atomic_fetch_add(sync_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      int guard = 1 << 18;                                    // (bounded: a lost count ends as a wrong answer the tests see, not a hang)
      while (__hip_atomic_load(sync_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)(sync_no * NP) && --guard > 0)
        __builtin_amdgcn_s_sleep(1);                          // (polling without the sleep: the same layer time)
      asm volatile("" ::: "memory");
    } else {
      __builtin_amdgcn_s_barrier();
    }
  };
  constexpr int NSYNC = 2;                                    // counters the quantiser passes: rows staged (+ sum of squares), codes written
  // roles + RMSNorm: the rows are requested by the four waves that run the sum of squares -- two chunks per thread, the tree's own
  // (chunk t8, chunk t8 + 256) -- so the sum runs on REGISTERS as soon as they arrive and one counter covers "rows staged" and "sum of
  // squares written" (a counter costs 400-700 cycles: LDS atomic + poll; profiles/r06/gemvq_trace_fine.txt)
  const bool regsum = NORM && roles;
  const int PR = regsum ? 256 : PT;                           // threads that request / stage the token rows

  // ---- everything the quantiser reads, requested first: one memory round trip for the whole prologue.  Whole waves without work
  // issue nothing (wave-uniform guards; a CU's vector-memory path takes ~20 cycles per wave instruction: the s_memtime trace of the
  // first version, profiles/r06/gemvq_trace.txt, had 4-8 k cycles of request issue in front of the quantiser)
  const int q_nchunks = H >> 3;                              // 16-byte chunks of a row
  const int tpr = H / CPT;                                   // tasks per row: 8 channels each, 16 per quantisation group
  const int ntask = p.M * tpr;
  const int wbase = tid & ~63;                               // this wave's first thread
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  constexpr int TPT = TPT1 * MT;
  v4u q_ri[TPT], q_rb[QOP == 4 ? TPT : 1];                  // per task: 8 reorder indices (ops 1-3) / 8 gate and 8 up values (op 4)
  h8 q_xr[ROWS ? XC : 1], q_rr[QOP == 3 ? XC : 1], q_wr[NORM ? XC : 1];
  typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));
  typedef float v2f_u __attribute__((ext_vector_type(2), aligned(4)));
  v4f_u po[QOP == 5 ? PH : 1][2];                           // QOP 5: my chunk's 8 values of the first PH splits, and every split's (m, d)
  v2f_u pmd[QOP == 5 ? PS : 1];
  v2f_u pmy = v2f_u{-INFINITY, 0.f};                        // QOP 5: the (m, d) of split lane % 16 of my head -- one request; the 16 lanes of a head share them through LDS
  const float *pwp = nullptr;
#pragma unroll
  for (int t = 0; t < TPT; ++t) {
    q_ri[t] = v4u{0u, 0u, 0u, 0u};
    if constexpr (QOP == 4) q_rb[t] = v4u{0u, 0u, 0u, 0u};
    if (!streamer && wbase + t * PT < ntask) {               // (wave-uniform)
      const int task = min(tid + t * PT, ntask - 1), m = MT == 1 ? 0 : task / tpr, e0 = (task - m * tpr) * CPT;   // (MT == 1: no division)
      if constexpr (QOP == 4) {
        q_ri[t] = *reinterpret_cast<const v4u *>(p.q_x + (int64_t)m * H + e0);
        q_rb[t] = *reinterpret_cast<const v4u *>(p.q_x2 + (int64_t)m * H + e0);
      } else if (p.q_idx) {
        q_ri[t] = *reinterpret_cast<const v4u *>(p.q_idx + e0);
      }
    }
  }
  if (QOP == 5 && !streamer) {                              // (one chunk per thread: M x K_total <= 8192, gemvq_merge_fits; wave-uniform)
    const int c = min(tid, p.M * q_nchunks - 1), m = MT == 1 ? 0 : c / q_nchunks, cc = c - m * q_nchunks;
    // thread j of a head's 16 takes values 4 j .. 4 j + 3 and 64 + 4 j .. + 3 of it: each request of the 16 lanes is 256 contiguous bytes
    // (8 contiguous values per thread -- every request touching all four lines of the record -- measured the same: the 133 KB a
    // workgroup reads take their 2 k cycles at the CU's 64 bytes per clock either way, profiles/r06/ab_merge_in_o_proj4.txt)
    const float *wp = p.q_part + ((int64_t)(m * (H >> 7) + (cc >> 4)) * p.q_splits) * 130 + (cc & 15) * 4;
    pwp = wp;
#pragma unroll
    for (int sp = 0; sp < PS; ++sp) {
      const int sc = min(sp, p.q_splits - 1);
      if (sp < PH) {
        po[sp][0] = po[sp][1] = v4f_u{0.f, 0.f, 0.f, 0.f};
        if (MT > 1 || sp < p.q_splits) {                    // (workgroup-uniform: a layer at context 1024 has 4 states -- 8 requests, not 16)
          po[sp][0] = *reinterpret_cast<const v4f_u *>(wp + sc * 130);
          po[sp][1] = *reinterpret_cast<const v4f_u *>(wp + sc * 130 + 64);
        }
      }
    }
    // (a wave's vector-memory instruction costs ~16 cycles of the CU's request path whatever its width: with every lane asking for all
    // 8-16 (m, d) pairs of its head the 33 requests per thread were issued at 5.0 k cycles, with one pair per lane at 4.3 k)
    pmy = *reinterpret_cast<const v2f_u *>(wp + min(lane & 15, p.q_splits - 1) * 130 + 128 - (cc & 15) * 4);
  }
  if constexpr (QOP <= 3) {
#pragma unroll
    for (int i = 0; i < XC; ++i) {
      q_xr[i] = h8{};
      if constexpr (QOP == 3) q_rr[i] = h8{};
      if (!streamer && tid < PR && wbase + i * PR < p.M * q_nchunks) {       // (wave-uniform: PR is a multiple of 64)
        const int c = min(tid + i * PR, p.M * q_nchunks - 1), m = MT == 1 ? 0 : c / q_nchunks, cc = c - m * q_nchunks;
        q_xr[i] = *reinterpret_cast<const h8 *>(reinterpret_cast<const char *>(p.q_x + (int64_t)m * H) + cc * 16);
        if constexpr (QOP == 3) q_rr[i] = *reinterpret_cast<const h8 *>(reinterpret_cast<const char *>(p.q_res + (int64_t)m * H) + cc * 16);
      }
    }
    if constexpr (QOP >= 2) {
#pragma unroll
      for (int i = 0; i < XC; ++i) {
        q_wr[i] = h8{};
        if (!streamer && wbase + i * PT < q_nchunks)
          q_wr[i] = *reinterpret_cast<const h8 *>(reinterpret_cast<const char *>(p.q_x2) + min(tid + i * PT, q_nchunks - 1) * 16);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (roles && tid == 0) *sync_cnt = 0u;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // every wave's quantiser requests are in the memory pipeline before ANY weight request goes out (one s_barrier: the 128 weight
  // requests of a workgroup would otherwise sit in front of the later waves' row chunks)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // ---- the weights of this wave's first D steps (NCH <= 2: its first D features)
  constexpr int PARTS = NCH > 6 ? 4 : (NCH > 2 ? 2 : 1), PCH = NCH / PARTS;
  // D = ring slots (template; the host's choice, launch1: a wave's steps up to two, HALF of them -- two or three slots -- beyond).  Every
  // slot is requested unconditionally and the first trip below is straight-line code, so that hipcc counts its waits (with requests
  // under run-time conditions it falls back to vmcnt(0) at every step: a wave then waited for its whole ring before its first feature and
  // ran its steps with nothing in flight, ~4 k idle cycles at the end of gate / up, profiles/r06/gemvq_trace.txt).  Where the two roles'
  // paths join in front of the feature loop the wait IS vmcnt(0) -- the ring was requested in different places --: one more reason for
  // a ring that is not the whole share.
  static_assert(PCH * PARTS == NCH && D % PARTS == 0, "chunks per lane: 1, 2, 4, 6 or 8");
  PartW<PCH> ring[D];
  v4i ring8[D / PARTS];                                           // the keeper chunks / scales of the features in the ring
  unsigned short ringsb8[D / PARTS];
  constexpr bool ADD = MT == 1 && D <= 4;                         // the output addend rides in the ring (one token, up to four slots: registers)
  unsigned short ringadd[ADD ? D / PARTS : 1];
  // (roles: the quantiser waves request their own features behind the last counter.  Giving ALL features to the streamers -- nothing
  // requested late -- measured the same: eight waves then do the arithmetic of sixteen, 1.2 k instead of 0.8 k cycles per feature step)
  // (computed HERE, behind the quantiser's requests, and with one 32-bit division: the 64-bit form cost ~250 scalar instructions in
  // front of the first request)
  const int fq = p.N / (int)gridDim.x, fr = p.N - fq * (int)gridDim.x;
  const int f0 = lw * fq + min(lw, fr), f1 = f0 + fq + (lw < fr ? 1 : 0);
  const bool sall = OWN && roles;                       // (OWN is only launched with roles)
  constexpr int fstride = OWN ? NWV - NP : NWV;
  const int n0w = f0 + (OWN ? wave - NP : wave);
  const int nfeat = (OWN && wave < NP) ? 0 : (n0w < f1 ? (f1 - n0w + fstride - 1) / fstride : 0);    // features of this wave
  const int nsteps = nfeat * PARTS;
  auto issue_ring = [&]() {
#pragma unroll
    for (int u = 0; u < D; ++u) {                         // (a slot past the wave's last step re-requests that step's feature: an L1 hit)
      const int fi = min(u / PARTS, max(nfeat - 1, 0));
      load_part<PCH, ADD>(p, min(n0w + fi * fstride, p.N - 1), u % PARTS, u % PARTS == PARTS - 1, lane, nchunks, ring[u], ring8[u / PARTS],
                          ringsb8[u / PARTS], &ringadd[ADD ? u / PARTS : 0]);
      if constexpr (MT == 1 && (QOP != 5 || OWN))         // step by step: hipcc otherwise issues every step's 16-byte loads first and the
        __builtin_amdgcn_sched_barrier(0);                // scale loads last -- and step 0 then waits for (nearly) the whole ring
    }                                                     // (two tokens / the merge op: no registers to spare for the fixed order)
  };
  // ---- the quantiser: the token rows' packed operand, built in LDS (its barriers wait for LDS only)
  uint8_t *qa4 = reinterpret_cast<uint8_t *>(lds);                                        // [MQ][K4h]
  uint8_t *qa8 = qa4 + MQ * K4h;                                                          // [MQ][128]
  half_t *qsa = reinterpret_cast<half_t *>(qa8 + MQ * kKeeper);                           // [G][MQ]
  half_t *qsa8 = qsa + (size_t)G * MQ;                                                    // [MQ]
  if (!streamer) {
  GQ_STAMP(1);                                               // every request issued
  __builtin_amdgcn_sched_barrier(0);
  // nothing of the quantiser moves up between the loads above, and none of its requests sinks into a branch below
#pragma unroll
  for (int t = 0; t < TPT; ++t) {
    asm volatile("" : "+v"(q_ri[t]));
    if constexpr (QOP == 4) asm volatile("" : "+v"(q_rb[t]));
  }
  if constexpr (QOP == 5) {
#pragma unroll
    for (int sp = 0; sp < PH; ++sp) asm volatile("" : "+v"(po[sp][0]), "+v"(po[sp][1]));
    asm volatile("" : "+v"(pmy));
  }
  if constexpr (QOP <= 3) {
#pragma unroll
    for (int i = 0; i < XC; ++i) {
      asm volatile("" : "+v"(q_xr[i]));
      if constexpr (QOP == 3) asm volatile("" : "+v"(q_rr[i]));
      if constexpr (QOP >= 2) asm volatile("" : "+v"(q_wr[i]));
    }
  }

    float *red = reinterpret_cast<float *>(lds + red_offset(K4h, G));                     // [MQ][4] partial sums of squares
    char *rowbuf = reinterpret_cast<char *>(red) + RED_BYTES;                             // [MQ][H] halves, then the norm weights [H]
    char *wbuf = rowbuf + MQ * H * 2;
    const int Gt = H >> 7;
    if constexpr (QOP == 5) {
      // out[dim] = sum_s o_s[dim] 2^(m_s - M) / sum_s d_s 2^(m_s - M), splits in order: decode_merge_kernel's operations, same bits
      {  // the head's (m, d) pairs: lane j of a head's 16 holds split j's; through the (unused) norm-weight area of LDS, wave-local --
         // a wave's LDS operations complete in order, no barrier
        v2f_u *mdx = reinterpret_cast<v2f_u *>(wbuf);
        if (tid < p.M * q_nchunks) mdx[tid] = pmy;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int sp = 0; sp < PS; ++sp) {
          pmd[sp] = v2f_u{-INFINITY, 0.f};
          if ((sp < PH || p.q_splits > PH) && tid < p.M * q_nchunks) pmd[sp] = mdx[(tid & ~15) + sp];
        }
      }
      float M_ = -INFINITY;
#pragma unroll
      for (int sp = 0; sp < PS; ++sp)
        if (sp < p.q_splits) M_ = fmaxf(M_, pmd[sp][0]);
      float acc[8], den = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
      for (int hb = 0; hb < PS / PH; ++hb) {
        if (hb > 0) {
          if (p.q_splits <= hb * PH) break;                  // (workgroup-uniform)
#pragma unroll
          for (int sp = 0; sp < PH; ++sp) {                  // the next batch of splits: one more trip to memory
            const int sc = min(hb * PH + sp, p.q_splits - 1);
            po[sp][0] = *reinterpret_cast<const v4f_u *>(pwp + sc * 130);
            po[sp][1] = *reinterpret_cast<const v4f_u *>(pwp + sc * 130 + 64);
          }
        }
#pragma unroll
        for (int sp = 0; sp < PH; ++sp)
          if (hb * PH + sp < p.q_splits) {
            const v2f_u md = pmd[hb * PH + sp];
            const float w = md[0] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(md[0] - M_);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_fmaf(po[sp][k >> 2][k & 3], w, acc[k]);
            den = __builtin_fmaf(md[1], w, den);
          }
      }
      if (tid < p.M * q_nchunks) {
        h8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (half_t)(den > 0.f ? acc[k] / den : 0.f);
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        char *hb = rowbuf + (tid & ~15) * 16 + (tid & 15) * 8;     // my head's 128 halves of the [M][H] rows: values 4 j .. and 64 + 4 j ..
        *reinterpret_cast<h4 *>(hb) = h4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<h4 *>(hb + 128) = h4{v[4], v[5], v[6], v[7]};
      }
    }
    if constexpr (QOP <= 3) {
#pragma unroll
      for (int i = 0; i < XC; ++i) {                         // rows (3: x + residual, one fp16 add per element as torch adds halves;
        const int c = tid + i * PR;                          // the first workgroup writes the residual stream)
        if (tid < PR && c < p.M * q_nchunks) {
          const int m = MT == 1 ? 0 : c / q_nchunks, cc = c - m * q_nchunks;
          h8 v = q_xr[i];
          if constexpr (QOP == 3) {
            v = v + q_rr[i];
            if (lw == 0) *reinterpret_cast<h8 *>(reinterpret_cast<char *>(p.q_res_out + (int64_t)m * H) + cc * 16) = v;
          }
          *reinterpret_cast<h8 *>(rowbuf + m * H * 2 + cc * 16) = v;
          q_xr[i] = v;                                       // (regsum: the sum of squares below reads it from here)
        }
      }
      if constexpr (QOP >= 2) {
#pragma unroll
        for (int i = 0; i < XC; ++i)
          if (tid + i * PT < q_nchunks) *reinterpret_cast<h8 *>(wbuf + (tid + i * PT) * 16) = q_wr[i];
      }
    }
    const bool sumsq_wave = NORM && (tid >> 8) < p.M;        // (wave-uniform) waves 4 m .. 4 m + 3: the tree of row m
    if constexpr (ROWS) {
      GQ_STAMP(2);                                             // the quantiser's inputs have arrived and sit in LDS
      if (!regsum) qsync();
      GQ_STAMP(3);
    }
    // Without roles the weights of this wave's first D steps go out HERE: in front of the quantiser's own requests they held every wave
    // at the CU's vector-memory issue rate (4-12 k cycles before the first row chunk reached LDS, profiles/r06/gemvq_trace.txt); from
    // here they overlap the sum of squares and the codes -- except on the waves that run the sum of squares (theirs go out behind it).
    // (Half of the ring up front, half here: measured equal to worse.  SiLU x up has no barrier in front of its codes and two 8-byte
    // inputs per task in registers: its ring goes out behind the codes.)
    if constexpr (QOP != 4) {
      if (!roles && !sumsq_wave) issue_ring();
    } else {
      if (p.q_roles & 2) issue_ring();                   // (SiLU x up, "early ring": behind the gate / up requests, in front of the codes)
    }
    float rinv[MQ] = {0.f, 0.f};
    if constexpr (QOP == 2 || QOP == 3) {
      const int m = tid >> 8, t8 = tid & 255;               // the stand-alone kernel's 4-wave tree, one per row
      if (m < p.M) {
        float ss = 0.f;
        if (regsum) {                                        // (one token: thread t8 holds chunks t8 and t8 + 256 = the tree's own)
#pragma unroll
          for (int i = 0; i < XC; ++i)
            if (t8 + i * 256 < q_nchunks) {
#pragma unroll
              for (int k = 0; k < 8; ++k) ss = __builtin_fmaf((float)q_xr[i][k], (float)q_xr[i][k], ss);
            }
        } else {
          for (int c = t8; c < q_nchunks; c += 256) {        // chunk (i * 4 + wave) * 64 + lane, i ascending
            const h8 v = *reinterpret_cast<const h8 *>(rowbuf + m * H * 2 + c * 16);
#pragma unroll
            for (int k = 0; k < 8; ++k) ss = __builtin_fmaf((float)v[k], (float)v[k], ss);
          }
        }
        ss = wave_sum_butterfly(ss);
        if (lane == 0) red[m * 4 + (wave & 3)] = ss;
      }
      qsync();
      GQ_STAMP(4);                                             // sum of squares done
      if (!roles && sumsq_wave) issue_ring();
#pragma unroll
      for (int m2 = 0; m2 < MT; ++m2) {                      // (every thread: ~110 instructions per row)
        const float tot = ((red[m2 * 4 + 0] + red[m2 * 4 + 1]) + red[m2 * 4 + 2]) + red[m2 * 4 + 3];
        const float var = (H & (H - 1)) == 0 ? tot * (1.0f / (float)H) : tot / (float)H;
        rinv[m2] = rinv_sqrt_exact(var + p.q_eps);
      }
    }
    GQ_STAMP(12);                                              // 1 / sqrt done
    // the codes: 8 channels per thread, a 128-channel group = 16 adjacent lanes (the stand-alone kernels take 16 per thread: the same
    // values, the same maximum, the same scale, the same codes -- spread over twice the threads; 4 per thread took two passes per wave
    // at hidden 4096, and a pass costs its VALU instructions -- two quantiser waves fill a SIMD: profiles/r06/gemvq_trace_fine.txt)
#pragma unroll
    for (int t = 0; t < TPT; ++t) {
      if (wbase + t * PT < ntask) {                          // (wave-uniform; ntask is a multiple of 16: whole groups per DPP row)
        const int task = min(tid + t * PT, ntask - 1);
        const int m = MT == 1 ? 0 : task / tpr, e0 = (task - m * tpr) * CPT;
        const int g = e0 >> 7, j = (e0 >> 3) & 15;
        const bool keeper = g == Gt - 1;
        float v[CPT];
        if constexpr (QOP == 4) {
          const half_t *av = reinterpret_cast<const half_t *>(&q_ri[t]), *bv = reinterpret_cast<const half_t *>(&q_rb[t]);
#pragma unroll
          for (int k = 0; k < CPT; ++k) v[k] = silu_mul<false>((float)av[k], (float)bv[k]);
        } else {
          const uint16_t *iv = reinterpret_cast<const uint16_t *>(&q_ri[t]);
          const float rv = m == 0 ? rinv[0] : rinv[1];
#pragma unroll
          for (int k = 0; k < CPT; ++k) {
            const int off = p.q_idx ? (int)iv[k] : e0 + k;
            const half_t xh = *reinterpret_cast<const half_t *>(rowbuf + m * H * 2 + off * 2);
            if constexpr (NORM) {
              const half_t wg = *reinterpret_cast<const half_t *>(wbuf + off * 2);
              v[k] = (float)(half_t)(((float)xh * (float)wg) * rv);                    // RMSNorm.cuh:145-151
            } else {
              v[k] = (float)xh;
            }
          }
        }
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) amax = fmaxf(amax, fabsf(v[i]));
        amax = max16(amax);
        const GroupScale gs = group_scale<false>(amax, keeper, p.q_clip);
        float tr[CPT];
#pragma unroll
        for (int i = 0; i < CPT; ++i) tr[i] = group_code<false>(v[i], gs);
        if (tid + t * PT < ntask) {
          if (keeper) {                                        // 8 INT8 codes: two words of pack_codes16's keeper form
            unsigned w2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const float lo = __builtin_fmaf(tr[4 * k + 1], 256.f, tr[4 * k] + 32896.f), hi = __builtin_fmaf(tr[4 * k + 3], 256.f, tr[4 * k + 2] + 32896.f);
              w2[k] = ((unsigned)lo | ((unsigned)hi << 16)) ^ 0x80808080u;
            }
            *reinterpret_cast<v2u *>(qa8 + m * kKeeper + j * 8) = v2u{w2[0], w2[1]};
          } else {                                             // 8 INT4 codes: one word of its nibble form
            float lo = 34952.f, hi = 34952.f;
            lo = __builtin_fmaf(tr[0], 1.f, lo);
            lo = __builtin_fmaf(tr[1], 16.f, lo);
            lo = __builtin_fmaf(tr[2], 256.f, lo);
            lo = __builtin_fmaf(tr[3], 4096.f, lo);
            hi = __builtin_fmaf(tr[4], 1.f, hi);
            hi = __builtin_fmaf(tr[5], 16.f, hi);
            hi = __builtin_fmaf(tr[6], 256.f, hi);
            hi = __builtin_fmaf(tr[7], 4096.f, hi);
            *reinterpret_cast<unsigned *>(qa4 + m * K4h + g * 64 + j * 4) = ((unsigned)lo | ((unsigned)hi << 16)) ^ 0x88888888u;
          }
          if (j == 0) {
            if (keeper) qsa8[m] = f2h(gs.s_store);
            else qsa[g * MQ + m] = f2h(gs.s_store);
          }
        }
      }
      if (t == 0) GQ_STAMP(13);                                // first pass of the codes done
    }
    if constexpr (QOP == 4) {
      if (!(p.q_roles & 2)) issue_ring();
    }
    GQ_STAMP(5);                                               // codes written
    qsync();
    GQ_STAMP(6);                                               // the packed operand is published
    if (roles && !sall) issue_ring();                      // the quantiser waves' own features
  } else {
    // streamers: the stream starts here; the operand is published once the quantiser's waves have passed their last counter
    issue_ring();
    GQ_STAMP(1);
    int guard = 1 << 20;
    while (__hip_atomic_load(sync_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)(NSYNC * NP) && --guard > 0)
      __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
    GQ_STAMP(6);
  }

  // ---- the feature loop: gemv1_w4a4_kernel's arithmetic, feature by feature.  Step s = (feature s / PARTS, part s % PARTS) sits in
  // ring slot s % D; a slot is re-filled with step s + D as soon as step s is computed (D - 1 steps of weights in flight per wave, no
  // register copies: the loop is unrolled by D, and PARTS | D keeps a feature's parts inside one trip)
  const bool leader = (lane & 3) == 0;
  // One token, rows of at most two chunks per lane, a ring of at most four slots (registers): the token's codes and scales this lane
  // multiplies with are the same for every feature -- read out of LDS ONCE behind the publication instead of at the head of every
  // feature step (two LDS round trips in front of each step's first dot product)
  constexpr bool HOIST = MT == 1 && NCH <= 2 && D <= 4 && (QOP != 5 || OWN);
  v4i ha[HOIST ? NCH : 1], ha8 = {};
  float hs[HOIST ? NCH : 1], hs8 = 0.f;
  if constexpr (HOIST) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int cc = min(lane + 64 * c, nchunks - 1);
      ha[c] = *reinterpret_cast<const v4i *>(qa4 + cc * 16);
      hs[c] = (float)qsa[(cc >> 2) * MQ];
    }
    ha8 = *reinterpret_cast<const v4i *>(qa8 + (lane & 7) * 16);
    hs8 = (float)qsa8[0];
  }
  auto finish = [&](int seg, int nl, const float (&acc)[MT], const v4i &w8, unsigned short sb8u, unsigned short addu) {   // keeper, 64-lane sum, output: gemv1_w4a4_kernel's tail
    const float sb8f = (float)__builtin_bit_cast(half_t, sb8u);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int mr = min(m, p.M - 1);
      const v4i a8 = HOIST ? ha8 : *reinterpret_cast<const v4i *>(qa8 + mr * kKeeper + (lane & 7) * 16);
      const float sa8f = HOIST ? hs8 : (float)qsa8[mr];
      int d = 0;
      d = __builtin_amdgcn_sdot4(a8[0], w8[0], d, false);
      d = __builtin_amdgcn_sdot4(a8[1], w8[1], d, false);
      d = __builtin_amdgcn_sdot4(a8[2], w8[2], d, false);
      d = __builtin_amdgcn_sdot4(a8[3], w8[3], d, false);
      d = quad_sum(d);
      d += __builtin_amdgcn_update_dpp(0, d, 0x104, 0xF, 0xF, true);   // row_shl:4 -- lane 0 += lane 4 (the other lanes' values are not used)
      float s = acc[m];
      s = wave_sum_butterfly(s);
      if (lane == 0 && m < p.M) {
        const float c = __builtin_fmaf((float)d, sa8f * sb8f, s);
        void *out = seg == 0 ? p.seg_out[0] : (seg == 1 ? p.seg_out[1] : p.seg_out[2]);
        const int64_t at = (int64_t)m * p.seg_n + nl;
        if ((p.seg_f32 >> seg) & 1u) {
          reinterpret_cast<float *>(out)[at] = c;
        } else {
          half_t h = f2h(c);
          if (seg == 0 && p.seg_add)                           // fp16 + fp16 as torch adds halves
            h = f2h((float)h + (ADD ? (float)__builtin_bit_cast(half_t, addu) : (float)p.seg_add[at]));
          reinterpret_cast<half_t *>(out)[at] = h;
        }
      }
    }
  };
  auto chunk = [&](int ch, const v4i &w, unsigned short sbu, float (&acc)[MT], int hc = 0) {   // one weight chunk against every token's codes (hc: the chunk's index per lane, HOIST)
    const bool ok = ch < nchunks;
    const int cc = min(ch, nchunks - 1);
    const float sbf = (float)__builtin_bit_cast(half_t, sbu);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int mr = min(m, p.M - 1);
      const v4i a = HOIST ? ha[HOIST ? hc : 0] : *reinterpret_cast<const v4i *>(qa4 + mr * K4h + cc * 16);
      const float saf = HOIST ? hs[HOIST ? hc : 0] : (float)qsa[(cc >> 2) * MQ + mr];
      int d = 0;
      d = __builtin_amdgcn_sdot8(a[0], w[0], d, false);
      d = __builtin_amdgcn_sdot8(a[1], w[1], d, false);
      d = __builtin_amdgcn_sdot8(a[2], w[2], d, false);
      d = __builtin_amdgcn_sdot8(a[3], w[3], d, false);
      d = quad_sum(d);                                    // exact: the group's 128-element integer dot
      const float next = __builtin_fmaf((float)d, saf * sbf, acc[m]);   // exact scale product
      acc[m] = (leader && ok) ? next : acc[m];
    }
  };
  // output segment of this wave's current feature, advanced by the feature stride (a division per feature cost ~20 scalar instructions
  // of the ~100 a step has)
  int seg = n0w / p.seg_n, nl = n0w - seg * p.seg_n;
  float acc[MT];
  auto trip = [&](auto first_c, int base) {
    constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int part = u % PARTS;
      const int fi = base / PARTS + u / PARTS;              // this wave's fi-th feature
      if (base + u < nsteps) {                              // (wave-uniform)
        const int n = n0w + fi * fstride;
        PartW<PCH> &f = ring[u];
        if (part == 0) {
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < PCH; ++c) chunk(lane + 64 * (part * PCH + c), f.w[c], f.sbu[c], acc, part * PCH + c);
        if (part == PARTS - 1) {
          finish(seg, nl, acc, ring8[u / PARTS], ringsb8[u / PARTS], ringadd[ADD ? u / PARTS : 0]);
          nl += fstride;
          while (nl >= p.seg_n) { nl -= p.seg_n; ++seg; }
        }
        if (FIRST && u < 4) GQ_STAMP(7 + u);                // steps 0 .. 3 done
        if (base + u + D < nsteps)                          // this slot's next tenant: step s + D = the same part of feature fi + D / PARTS
          load_part<PCH, ADD>(p, n + (D / PARTS) * fstride, part, part == PARTS - 1, lane, nchunks, f, ring8[u / PARTS], ringsb8[u / PARTS],
                              &ringadd[ADD ? u / PARTS : 0]);
      }
    }
  };
  if constexpr (MT == 1) {
    trip(std::true_type{}, 0);                               // straight-line: precise waits on the ring
    for (int base = D; base < nsteps; base += D) trip(std::false_type{}, base);
  } else {                                                   // (two tokens: one copy of the step code -- registers)
    for (int base = 0; base < nsteps; base += D) trip(std::false_type{}, base);
  }
  GQ_STAMP(11);                                              // all features of this wave done
#ifdef ATOM_TOOLS
  if (trb && lane == 0) trb[14] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
}

template <int QOP, int NCH, int MT, int D, bool OWN = false>
static int launch1d(const GemmParams &p, hipStream_t s) {
  // roles (the kernel's comment): one token, ops 1-3, the rows and the norm weight within one 16-byte chunk per thread of the eight
  // quantiser waves and at most TPT1 tasks per thread of them.  (Round 6 also built LOADER waves that fetched every wave's first
  // features into LDS by LDS-DMA: bit-identical and slower -- profiles/r06/ab_gemvq_lds_prefetch.txt, commit ea86eed -- and removed.)
  GemmParams q = p;
  const int H_ = p.K4h * 2 + kKeeper;
  q.q_roles = ((QOP <= 3 || QOP == 5) && MT == 1 && p.M == 1 && H_ / 8 <= 8 * 64 && H_ / CPT <= TPT1 * 8 * 64) ? ATOM_TUNE("ATOM_GEMVQ_ROLES", 1) : 0;
  if (QOP == 4) q.q_roles = ATOM_TUNE("ATOM_GEMVQ_EARLY4", 1) ? 2 : 0;   // bit 1: SiLU x up requests its weight ring in front of the codes
  const size_t lds = lds_bytes(QOP, p.K4h, p.G);
  static std::atomic<uint64_t> attr_done{0};
  if (ensure_max_lds(reinterpret_cast<const void *>(&gemvq_w4a4_kernel<QOP, NCH, MT, D, OWN>), 128 * 1024, attr_done) != ATOM_OK) return ATOM_ERR_LAUNCH;
  // one workgroup per CU at most; every wave of it at least one feature
  const int cap = ATOM_TUNE("ATOM_GEMVQ_GRID", 256);
  int grid = p.N / NWV;
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((gemvq_w4a4_kernel<QOP, NCH, MT, D, OWN>), dim3((unsigned)grid), dim3(NTH), lds, s, q);
  return check_launch();
}

// ring depth by the steps per wave of this launch (its features x parts): see the rule below; 4 slots for rows of more than 4 chunks per
// lane and at two tokens (registers)
template <int QOP, int NCH, int MT>
static int launch1(const GemmParams &p, hipStream_t s) {
  constexpr int PARTS = NCH > 6 ? 4 : (NCH > 2 ? 2 : 1);
  int grid = p.N / NWV;
  if (grid > 256) grid = 256;
  if (grid < 1) grid = 1;
  const int per_wg = (p.N + grid - 1) / grid, steps = (per_wg + NWV - 1) / NWV * PARTS;
  if constexpr (MT == 1 && NCH <= 2 && (QOP <= 3 || QOP == 5)) {
    // roles (launch1d's predicate) and a share of at most two steps per streamer wave: they own every feature (OWN)
    const int H_ = p.K4h * 2 + kKeeper, sall = (per_wg + 7) / 8;
    if (p.M == 1 && H_ / 8 <= 8 * 64 && H_ / CPT <= TPT1 * 8 * 64 && ATOM_TUNE("ATOM_GEMVQ_ROLES", 1) && ATOM_TUNE("ATOM_GEMVQ_OWN", 1) && sall <= 2)
      return (sall <= 1 || ATOM_TUNE("ATOM_GEMVQ_OWN_D", 2) == 1) ? launch1d<QOP, NCH, MT, 1, true>(p, s) : launch1d<QOP, NCH, MT, 2, true>(p, s);
  }
  if constexpr (MT == 1 && NCH <= 4) {
    if constexpr (PARTS == 1) {
      // HALF a wave's share in the ring from three steps on (q / k / v: 3 steps, 2 slots; gate / up: 6 steps, 3 slots).  The first form
      // of this rule put the WHOLE share in flight (ring = steps, up to six slots): that was right while conditional refills made hipcc
      // wait for the whole ring at every step; with exact waits it is wrong -- a wave cannot compute while it sits in request issue, and
      // a CU hands out ~10 bytes per clock: the streamers of gate / up were still ISSUING their six slots at 16.7 k cycles, 8 k after
      // the operand was published, and ran their six steps behind that (profiles/r06/gemvq_trace_final.txt).  Same box, us per layer
      // cold: six slots 45.7, three 44.1, two 43.8-44.0, one 45.2; per kernel gate / up 12.9 / 11.3 / 11.7 (6 / 3 / 2 slots),
      // q / k / v 8.56 / 8.01 (3 / 2): profiles/r06/ab_gemvq_ring_depth2.txt.  (ATOM_GEMVQ_DMAX = n in tuning builds: min(steps, n).)
      const int dmax = ATOM_TUNE("ATOM_GEMVQ_DMAX", 0);
      const int d = dmax > 0 ? (steps < dmax ? steps : dmax) : (steps <= 2 ? steps : (steps <= 4 ? 2 : 3));
      if (d <= 1) return launch1d<QOP, NCH, MT, 1>(p, s);
      if (d == 2) return launch1d<QOP, NCH, MT, 2>(p, s);
      if (d == 3) return launch1d<QOP, NCH, MT, 3>(p, s);
      if (d <= 5 || QOP == 5) return launch1d<QOP, NCH, MT, 4>(p, s);   // (the merge op has no registers for six slots)
      return launch1d<QOP, NCH, MT, QOP == 5 ? 4 : 6>(p, s);
    } else {
      // rows of 3-4 chunks per lane (hidden 5120: Llama-13B): a feature = two steps, and ONE feature in the ring -- the same finding as
      // above, every wave quantises and streams here (no roles: the row needs more than eight waves' threads): a 13B layer at batch 1
      // 73.6 (six steps in the ring) / 73.5 (four) / 72.0 us (two), profiles/r06/ab_gemvq_ring_depth_13b.txt.  (ATOM_GEMVQ_DMAX2: tuning)
      const int dmax = ATOM_TUNE("ATOM_GEMVQ_DMAX2", 2);
      if (steps <= 2 || dmax <= 2) return launch1d<QOP, NCH, MT, 2>(p, s);
      if (steps <= 4 || QOP == 5 || dmax <= 4) return launch1d<QOP, NCH, MT, 4>(p, s);
      return launch1d<QOP, NCH, MT, QOP == 5 ? 4 : 6>(p, s);
    }
  } else {
    return launch1d<QOP, NCH, MT, 4>(p, s);
  }
}

template <int QOP, int MT>
static int launch_nch(const GemmParams &p, hipStream_t s) {
  const int need = ((p.K4h >> 4) + 63) / 64;
  if (need <= 2) return launch1<QOP, 2, MT>(p, s);
  if (need <= 4) return launch1<QOP, 4, MT>(p, s);
  // (5 .. 8 chunks per lane run the 8-chunk instance -- four parts of two: its ring is 40 registers where three-chunk parts need 56 and
  // spill; the chunks past the row's end are one clamped address per wave)
  if (need <= 8) return launch1<QOP, 8, MT>(p, s);
  return ATOM_ERR_SHAPE;
}

template <int QOP>
static int launch_mt(const GemmParams &p, hipStream_t s) {
  return p.M <= 1 ? launch_nch<QOP, 1>(p, s) : launch_nch<QOP, 2>(p, s);
}

}  // namespace gemvq

// THE shape predicate of this launch (atom_gemm_w4a4_multi_q routes one or two tokens here when it holds): per thread of the 1024 at
// most two quantiser tasks of 8 channels per token row -- kept at hidden <= 12,288, the range the tests cover -- and two 16-byte chunks of
// the token rows / the norm weight (ops 1-3), at most 8 weight chunks per lane (K_total <= 16,512), everything within 128 KiB of LDS.
bool gemvq_fits(int q_op, int64_t M, int64_t N, int64_t H) {
  if (q_op < 1 || q_op > 4 || M < 1 || M > gemvq::MQ || H < 2 * kKeeper || ((H - kKeeper) % kGroup) != 0 || N < gemvq::NWV) return false;
  if (H / gemvq::CPT > gemvq::TPT1 * gemvq::NTH || H > 12288) return false;
  if (q_op <= 3 && (M * (H >> 3) > gemvq::XC * gemvq::NTH || (H >> 3) > gemvq::XC * gemvq::NTH)) return false;
  const int K4h = (int)((H - kKeeper) / 2), G = (int)((H - kKeeper) / kGroup);
  const int need = ((K4h >> 4) + 63) / 64;                 // weight chunks per lane
  if (need > 8) return false;
  return gemvq::lds_bytes(q_op, K4h, G) <= (size_t)128 * 1024;
}

// ... of the merge form (q_op 5): one 16-byte row chunk per thread (M x K_total <= 8192), at most 16 KV splits, K_total = heads x 128
bool gemvq_merge_fits(int64_t M, int64_t N, int64_t H, int splits) {
  if (splits < 2 || splits > 16 || M < 1 || M > gemvq::MQ || (H % 128) != 0 || M * (H >> 3) > gemvq::NTH) return false;
  return gemvq_fits(1, M, N, H);
}

int launch_gemvq_multi_q(const GemmParams &p, hipStream_t s) {
  if (p.seg_n < 1 || (p.N % p.seg_n) != 0 || p.N / p.seg_n > 3 || !p.seg_out[0]) return ATOM_ERR_SHAPE;
  if (p.q_op == 5) {
    if (!gemvq_merge_fits(p.M, p.N, 2 * (int64_t)p.K4h + kKeeper, p.q_splits) || !p.q_part) return ATOM_ERR_SHAPE;
    return gemvq::launch_mt<5>(p, s);
  }
  if (!gemvq_fits(p.q_op, p.M, p.N, 2 * (int64_t)p.K4h + kKeeper)) return ATOM_ERR_SHAPE;
  switch (p.q_op) {
    case 1: return gemvq::launch_mt<1>(p, s);
    case 2: return gemvq::launch_mt<2>(p, s);
    case 3: return gemvq::launch_mt<3>(p, s);
    case 4: return gemvq::launch_mt<4>(p, s);
  }
  return ATOM_ERR_INVALID_ARG;
}

}  // namespace atom
