// W4A4 GEMM, decode batches (2 <= M <= 256) for gfx950: weight streaming on the INT8 MFMA, latency-bound.
//
// A serving step multiplies a handful of tokens by the whole weight matrix: 8.4 MB of packed weights at N=K=4096 against
// 0.1-2 GFLOP.  The tile kernels need 64-row tiles and a K loop with a barrier per group (12-15 us through split-K and a
// second launch); the dot-product decode kernel (gemv_w4a4.hip) pays VALU and a cross-lane reduction per token.  Here:
//   * one workgroup owns 16 output features; its 8 waves split K (groups in order, keeper last), so a shape yields N/16
//     workgroups (256 at N=4096) and every wave has its whole weight slice in flight from its first instruction:
//     lane (feature l%16, chunk l/16) loads the 16-byte chunk of each of its <= 14 groups straight into VGPRs
//     (8 MB in flight over the chip; no LDS round trip, every weight byte is read exactly once);
//   * v_mfma_i32_16x16x64_i8: A = 16 features x 64 channels, B = 64 channels x 16 tokens.  A packed 16-byte chunk holds 32
//     codes: the even ones (low nibbles) as (x << 4) & 0xF0F0F0F0 and the odd ones as x & 0xF0F0F0F0 are int8 = 16*code;
//     the same permutation of k on both operands, so two MFMAs give 256 * (the group's integer dot), exactly;
//   * token blocks of 16 (up to 16) re-use the weight registers; a block's activation chunks come from L2 and are
//     re-filled in place for the next block as soon as they have been widened;
//   * per group and 16x16 tile: c = fma(idot, sA[m,g] * sB[g,n], c), the scale product exact -- the contract of
//     include/atom_hip.h; the partial sums of the 8 waves are added in wave order through LDS (the FP32 summation ORDER
//     therefore differs from the prefill kernels, like the decode kernel's; all are within 1 fp16 ulp of the exact value).
// Round 3, what did NOT help for 32..256 tokens (profiles/r03_decode.txt): every block of a workgroup requested up front instead of
// one block ahead, 64-token workgroups over blockIdx.y, weights widened / scales converted once per workgroup instead of per block,
// the magic-number de-quantisation (2 instead of 3 VALU per element): 64 tokens 5.9-6.7 us against 6.1, 256 tokens 15.4-22.5 against
// 15.0.  The limiter there is the activation volume -- every 16-feature workgroup re-reads M x K/2 bytes in 64-byte row pieces,
// 128 MB per launch at 256 tokens -- or so it seemed: a second version that staged each token block's rows (one contiguous G KiB slab)
// in LDS by LDS-DMA, double-buffered, one barrier per block, registers independent of M, measured 7.2 / 11.4 / 19.7 us at 64 / 128 /
// 256 tokens (this kernel: 6.2 / 9.1 / 15.4), and an ablation of THIS kernel at 256 tokens (tools build) reads: full 18.3 us, without
// the per-block activation loads 18.0, without MFMA + de-quantisation 12.8, without both 10.1, against 3.5 us for one block.  The
// block loop is bound by what a wave ISSUES per block (widening both operands, the scale loads, address arithmetic: ~170
// instructions at ~5.5 cycles each with two waves per SIMD), not by memory: the next step for 64+ tokens is a tile kernel whose
// operands arrive widened (the F6 K-group kernels already take over at 256 rows), not another variant of this loop.
// Nor is it instruction fetch of the unrolled copies: the token blocks as a run-time loop (one block's code, a block's sums parked
// in LDS as soon as it is done, 114 VGPRs) measured 7.1 / 11.0 / 18.7 us at 64 / 128 / 256 tokens against 6.4 / 9.5 / 15.5 (same box).
// Replaces the M = 16..256 rows of the reference's NVBench sweep (kernels/src/GEMM/bench_dense_layer_gemm_i4_o16.cu:64-69),
// which runs the 128x128 tensor-core tile kernel for every M (26.7-27.2 us on the RTX 4090).
#include <cstdlib>
#include "common.h"
#include "quant_math.h"

namespace atom {
namespace skinny {

constexpr int CNT_MAX = 14;                              // items (int4 groups, or the keeper) per wave, at most

__device__ __forceinline__ v4i even_codes(v4u x) {       // low nibbles  -> int8 16*code
  return v4i{(int)((x.x << 4) & 0xF0F0F0F0u), (int)((x.y << 4) & 0xF0F0F0F0u), (int)((x.z << 4) & 0xF0F0F0F0u),
             (int)((x.w << 4) & 0xF0F0F0F0u)};
}
__device__ __forceinline__ v4i odd_codes(v4u x) {        // high nibbles -> int8 16*code
  return v4i{(int)(x.x & 0xF0F0F0F0u), (int)(x.y & 0xF0F0F0F0u), (int)(x.z & 0xF0F0F0F0u), (int)(x.w & 0xF0F0F0F0u)};
}

// c[r] = fma(idot[r], sa * sb[r], c[r]), r = the lane's 4 features: the contract of include/atom_hip.h (round 5) -- the scale product
// is exact in FP32 (two fp16 values; the 1/256 of the widened operands folded into sa is a power of two), one rounding per group
__device__ __forceinline__ void dequant4(const v4i &acc, float sa, const v2u &sb, float (&c)[4]) {
  const half_t *hv = reinterpret_cast<const half_t *>(&sb);
  float t[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = (float)hv[r] * sa;
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = __builtin_fmaf((float)acc[r], t[r], c[r]);
}

// fused quantiser (QOP): byte offset of red[4] (then the row buffer) behind the packed operand of MQ = 2 rows in LDS
__host__ __device__ inline int q_red_offset(int K4h, int G) { return (2 * K4h + 2 * kKeeper + 2 * G * 2 + 2 * 2 + 15) & ~15; }
constexpr int kQWaves = 8;                 // the quantiser-in-front variant runs 8-wave workgroups only (launch_gemm_skinny_multi_q)
constexpr int kQLdsMax = 96 * 1024;
// LDS of a quantiser-in-front launch: the partial sums of the 8 waves, the packed operand built in place, the reduction scratch, and
// (ops 1-3) the fp16 rows of up to two tokens + the norm weights
inline size_t q_lds_bytes(int q_op, int K4h, int G) {
  const int H = 2 * K4h + kKeeper;
  return (size_t)2 * kQWaves * 64 * 16 + q_red_offset(K4h, G) + 32 + (q_op <= 3 ? (size_t)H * 2 * 3 : 0);
}

// NW waves per workgroup, MBLK token blocks of 16, CNT = register slots for the wave's items (>= ceil((G + 1) / NW))
// OUT: 0 = fp16 D [M, N]; 1 = FP32 sums to p.ws [M, N] (the u4-epilogue path); 2 = segmented (atom_gemm_w4a4_multi): the features are
// p.N / p.seg_n segments with their own [M, seg_n] outputs, fp16 or float32 per segment, segment 0 optionally + an fp16 addend
// NT: the weight loads are non-temporal (global_load ... nt).  Every weight byte is read once, by one CU; with the default policy the
// stream still allocates in the caches it passes.  For one or two tokens (profiles/r03_decode.txt item 6; HBM-cold, same box, us):
// 1 x 4096 x 4096 5.33 -> 4.95, 1 x 11008 x 4096 8.6 -> 8.3, the Llama-7B decode layer 70.7 -> 66.4 at batch 1 and 73.6 -> 70.1 at
// batch 2.  From 4 tokens the large shapes measure 3-8 % SLOWER with it (4 x 11008 x 4096 7.9 -> 8.4, 16 x 11008 x 4096 9.15 -> 9.53)
// and the layer is unchanged, so larger batches keep the default.  (A weight set that fits the 256 MB Infinity Cache and is replayed
// -- the "hot" columns -- loses its residency with nt: hot becomes cold.)
// QOP != 0 (one or two tokens, MBLK == 1; atom_gemm_w4a4_multi_q): the kernel starts with the quantiser that feeds this GEMM in the
// reference's call order -- reorder (1), RMSNorm + reorder (2), residual add + RMSNorm + reorder (3), SiLU x up (4); punica/models/
// llama.py:259-292, :85-87 -- run by EVERY workgroup on its own copy of the token rows, behind the weight loads (which are in flight
// while it runs), and its packed operand stays in LDS.  Same arithmetic as quant_kernels.hip slot by slot (kernel-flavoured mode:
// Reorder.cuh:137-178, RMSNorm.cuh:112-151, Activate.cuh:112-167) including the fixed-shape FP32 tree of the sum of squares (256
// threads, chunk c -> wave (c / 64) % 4, lane c % 64): bit-identical to the separate launch, which costs a launch boundary and a
// round trip through HBM more than the GEMM itself at this size.
template <int NW, int MBLK, int CNT, int OUT = 0, bool NT = false, int QOP = 0>
__global__ __launch_bounds__(NW * 64) void gemm_w4a4_skinny_kernel(GemmParams p) {
  constexpr bool OUT32 = OUT == 1;
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];      // float part[NW][MBLK][64][4] (QOP: two of them); QOP: + the quantiser's buffers
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane & 15, kb = lane >> 4;
  // XCD-aware feature map (round 4, as in gemv_w4a4.hip): workgroup b runs on XCD b % 8; a 64-byte line of weight scales holds 32
  // adjacent features of one group = two workgroups' worth -- dealt round-robin they sit in two L2s.  Bijective for any grid size.
  const int xq = (int)gridDim.x >> 3, xr = (int)gridDim.x & 7, xx = blockIdx.x & 7;
  const int lw = xx * xq + min(xx, xr) + ((int)blockIdx.x >> 3);      // workgroups in XCD-contiguous order
  // QOP (round 6): the grid may be SMALLER than the N / 16 feature blocks -- workgroup lw then owns the consecutive blocks
  // [lw NB / grid, (lw + 1) NB / grid) and runs its quantiser ONCE in front of all of them (one workgroup per block repeated it 768 times
  // for q / k / v, 1,376 times for gate / up: ~1 us of a CU's issue time each, which is what made those fusions lose in rounds 3-5);
  // the weight registers of block b + 1 are re-filled in place as block b consumes them.
  // (LOOP: instances of up to 8 register slots per wave -- K <= 8,064; the 14-slot instance has no registers to spare for the
  // refills and keeps one workgroup per block: launch_q1 sizes its grid accordingly)
  constexpr bool LOOP = QOP != 0 && CNT <= 8;
  const int NBLK = p.N >> 4;
  const int blk0 = LOOP ? (int)((int64_t)lw * NBLK / (int)gridDim.x) : lw;
  const int blk1 = LOOP ? (int)((int64_t)(lw + 1) * NBLK / (int)gridDim.x) : lw + 1;
  int n0 = blk0 * 16;
  const int K4h = p.K4h, G = p.G;

  // this wave's items: int4 groups [i0, min(i1, G)), and the keeper if i1 == G + 1
  const int per = (G + 1 + NW - 1) / NW;
  const int i0 = wave * per, i1 = min(i0 + per, G + 1);
  const bool keeper = i1 == G + 1 && i0 <= G;
  const int ng = min(i1, G) - i0;                          // int4 groups of this wave (<= CNT; may be <= 0)

  // addresses = wave-uniform base (SGPRs) + 32-bit lane offset (one VGPR) + compile-time j * 64 where the stride is known:
  // 64-bit per-lane pointers for 4 x 14 loads would cost more registers than the data
  const char *wbase = reinterpret_cast<const char *>(p.B4) + (int64_t)n0 * K4h + i0 * 64;
  const unsigned woff = (unsigned)(row * K4h + kb * 16);
  const char *sbbase = reinterpret_cast<const char *>(p.sB + (int64_t)i0 * p.N + n0);
  const char *abase = reinterpret_cast<const char *>(p.A4) + i0 * 64;
  const char *sabase = reinterpret_cast<const char *>(p.sA + (int64_t)i0 * p.ldA);
  unsigned aoff[MBLK], soff[MBLK], koff[MBLK];
#pragma unroll
  for (int tb = 0; tb < MBLK; ++tb) {
    const int m = min(tb * 16 + row, p.M - 1);
    aoff[tb] = (unsigned)(m * K4h + kb * 16);
    soff[tb] = (unsigned)(p.ref_layout ? ref_scale_index(m) : m) * 2u;
    koff[tb] = (unsigned)(m * kKeeper + kb * 16);
  }

  // ---- QOP: everything the fused quantiser reads is requested FIRST (one memory round trip for the whole prologue, and its waits
  // then count only the weight loads issued behind it): the thread's chunks of the token rows / residual rows / norm weights and the
  // reorder indices (4: the gate / up values) of its slot tasks
  constexpr int MQ = 2;                                     // rows the fused quantiser handles
  constexpr int TPT = QOP == 4 ? 3 : 2, XC = 3;             // slot tasks / row chunks per thread at most (checked by the launcher)
  const int H = 2 * K4h + kKeeper;
  const int q_nchunks = H >> 3, q_nslots = H >> 4;
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  v4u q_ri[QOP ? TPT : 1][2], q_rb[QOP == 4 ? TPT : 1][2];
  h8 q_xr[QOP && QOP <= 3 ? XC : 1], q_rr[QOP == 3 ? XC : 1], q_wr[QOP == 2 || QOP == 3 ? 2 : 1];
  if constexpr (QOP != 0) {
    const int tid = threadIdx.x, ntask = p.M * q_nslots;
#pragma unroll
    for (int t = 0; t < TPT; ++t) {
      const int task = min(tid + t * NW * 64, ntask - 1), m = task / q_nslots, e0 = (task - m * q_nslots) * 16;
      if constexpr (QOP == 4) {
        const half_t *arow = p.q_x + (int64_t)m * H, *brow = p.q_x2 + (int64_t)m * H;
        q_ri[t][0] = *reinterpret_cast<const v4u *>(arow + e0);
        q_ri[t][1] = *reinterpret_cast<const v4u *>(arow + e0 + 8);
        q_rb[t][0] = *reinterpret_cast<const v4u *>(brow + e0);
        q_rb[t][1] = *reinterpret_cast<const v4u *>(brow + e0 + 8);
      } else {
        const int16_t *ip = p.q_idx ? p.q_idx + e0 : reinterpret_cast<const int16_t *>(p.q_x);   // (no index: any readable address)
        q_ri[t][0] = *reinterpret_cast<const v4u *>(ip);
        q_ri[t][1] = *reinterpret_cast<const v4u *>(ip + 8);
      }
    }
    if constexpr (QOP <= 3) {
#pragma unroll
      for (int i = 0; i < XC; ++i) {
        const int c = min(tid + i * NW * 64, p.M * q_nchunks - 1), m = c / q_nchunks, cc = c - m * q_nchunks;
        q_xr[i] = *reinterpret_cast<const h8 *>(reinterpret_cast<const char *>(p.q_x + (int64_t)m * H) + cc * 16);
        if constexpr (QOP == 3) q_rr[i] = *reinterpret_cast<const h8 *>(reinterpret_cast<const char *>(p.q_res + (int64_t)m * H) + cc * 16);
      }
      if constexpr (QOP >= 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          q_wr[i] = *reinterpret_cast<const h8 *>(reinterpret_cast<const char *>(p.q_x2) + min(tid + i * NW * 64, q_nchunks - 1) * 16);
      }
    }
  }

  if constexpr (QOP != 0) __builtin_amdgcn_sched_barrier(0);   // (the requests above are issued before the weight loads below)
  // ---- everything this wave will ever read of the weights, in flight at once (QOP: the same number of loads in every wave --
  // slots past the wave's groups re-read its last one -- so that the waits above them are exact counts)
  v4u w[CNT];
  v2u sb[CNT];
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    if (QOP != 0 || j < ng) {
      const int jj = QOP != 0 ? max(min(j, ng - 1), -i0) : j;
      const v4u *wp = reinterpret_cast<const v4u *>(wbase + jj * 64 + woff);
      w[j] = NT ? __builtin_nontemporal_load(wp) : *wp;
      sb[j] = *reinterpret_cast<const v2u *>(sbbase + (int64_t)jj * p.N * 2 + 8 * kb);
    }
  }
  v4u wk[2] = {};
  v2u sbk = {};
  if (QOP != 0 || keeper) {
    const char *kp = reinterpret_cast<const char *>(p.B8) + (int64_t)n0 * kKeeper + (unsigned)(row * kKeeper + kb * 16);
    wk[0] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4u *>(kp)) : *reinterpret_cast<const v4u *>(kp);
    wk[1] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4u *>(kp + 64)) : *reinterpret_cast<const v4u *>(kp + 64);
    sbk = *reinterpret_cast<const v2u *>(reinterpret_cast<const char *>(p.sB8 + n0) + 8 * kb);
  }
  // ---- token block 0
  // ---- QOP: the token rows' packed operand, built in LDS while the weight loads are in flight (its barriers wait for LDS only)
  if constexpr (QOP != 0) {
    // nothing of the quantiser moves up between the loads above, and none of its requests sinks into a branch below (hipcc moves a
    // load whose only use is conditional into that branch -- behind the weight loads, with a full wait in front of its use)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TPT; ++t) {
      asm volatile("" : "+v"(q_ri[t][0]), "+v"(q_ri[t][1]));
      if constexpr (QOP == 4) asm volatile("" : "+v"(q_rb[t][0]), "+v"(q_rb[t][1]));
    }
    if constexpr (QOP <= 3) {
#pragma unroll
      for (int i = 0; i < XC; ++i) {
        asm volatile("" : "+v"(q_xr[i]));
        if constexpr (QOP == 3) asm volatile("" : "+v"(q_rr[i]));
      }
      if constexpr (QOP >= 2) asm volatile("" : "+v"(q_wr[0]), "+v"(q_wr[1]));
    }
  }
  char *qbase = lds_raw + (size_t)(QOP != 0 ? 2 : 1) * NW * MBLK * 64 * 16;      // behind the partial-sum area(s)
  uint8_t *qa4 = reinterpret_cast<uint8_t *>(qbase);                                     // [MQ][K4h]
  uint8_t *qa8 = qa4 + MQ * K4h;                                                        // [MQ][128]
  half_t *qsa = reinterpret_cast<half_t *>(qa8 + MQ * kKeeper);                         // [G][MQ]
  half_t *qsa8 = qsa + (size_t)G * MQ;                                                  // [MQ]; then, 16-byte aligned: red, rows, norm weights
  if constexpr (QOP != 0) {
    static_assert(MBLK == 1, "fused quantiser: one token block");
    auto lds_barrier = [] {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    const int tid = threadIdx.x;
    float *red = reinterpret_cast<float *>(qbase + q_red_offset(K4h, G));               // [MQ][4] partial sums of squares
    char *rowbuf = reinterpret_cast<char *>(red + 8);                                   // [MQ][H] halves, then the norm weights [H]
    char *wbuf = rowbuf + MQ * H * 2;
    const int nchunks = q_nchunks, nslots = q_nslots, Gt = H >> 7;
    const int ntask = p.M * nslots;                          // slot tasks: (row, slot), 16 channels each
    if constexpr (QOP <= 3) {
#pragma unroll
      for (int i = 0; i < XC; ++i) {                         // rows (3: x + residual, one fp16 add per element as torch adds halves;
        // workgroup 0 writes the residual stream)
        const int c = tid + i * NW * 64;
        if (c < p.M * nchunks) {
          const int m = c / nchunks, cc = c - m * nchunks;
          h8 v = q_xr[i];
          if constexpr (QOP == 3) {
            v = v + q_rr[i];
            if (lw == 0) *reinterpret_cast<h8 *>(reinterpret_cast<char *>(p.q_res_out + (int64_t)m * H) + cc * 16) = v;
          }
          *reinterpret_cast<h8 *>(rowbuf + m * H * 2 + cc * 16) = v;
        }
      }
      if constexpr (QOP >= 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (tid + i * NW * 64 < nchunks) *reinterpret_cast<h8 *>(wbuf + (tid + i * NW * 64) * 16) = q_wr[i];
      }
      lds_barrier();
    }
    float rinv[MQ] = {0.f, 0.f};
    if constexpr (QOP == 2 || QOP == 3) {
      const int m = tid >> 8, t8 = tid & 255;               // the stand-alone kernel's 4-wave tree, one per row
      if (m < p.M) {
        float ss = 0.f;
        for (int c = t8; c < nchunks; c += 256) {            // chunk (i * 4 + wave) * 64 + lane, i ascending
          const h8 v = *reinterpret_cast<const h8 *>(rowbuf + m * H * 2 + c * 16);
#pragma unroll
          for (int k = 0; k < 8; ++k) ss = __builtin_fmaf((float)v[k], (float)v[k], ss);
        }
        ss = wave_sum_butterfly(ss);
        if (lane == 0) red[m * 4 + (wave & 3)] = ss;
      }
      lds_barrier();
#pragma unroll
      for (int m2 = 0; m2 < MQ; ++m2) {
        const float tot = ((red[m2 * 4 + 0] + red[m2 * 4 + 1]) + red[m2 * 4 + 2]) + red[m2 * 4 + 3];
        const float var = (H & (H - 1)) == 0 ? tot * (1.0f / (float)H) : tot / (float)H;
        rinv[m2] = rinv_sqrt_exact(var + p.q_eps);
      }
    }
#pragma unroll
    for (int t = 0; t < TPT; ++t) {
      const int task = tid + t * NW * 64;
      if (task < ntask) {                                    // (ntask is a multiple of 8: whole octets for max8)
        const int m = task / nslots, slot = task - m * nslots;
        const int e0 = slot * 16, g = slot >> 3, j = slot & 7;
        const bool keeper = g == Gt - 1;
        float v[16];
        if constexpr (QOP == 4) {
          const half_t *av = reinterpret_cast<const half_t *>(q_ri[t]), *bv = reinterpret_cast<const half_t *>(q_rb[t]);
#pragma unroll
          for (int k = 0; k < 16; ++k) v[k] = silu_mul<false>((float)av[k], (float)bv[k]);
        } else {
          const uint16_t *iv = reinterpret_cast<const uint16_t *>(q_ri[t]);
          const float rv = m == 0 ? rinv[0] : rinv[1];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int off = p.q_idx ? (int)iv[k] : e0 + k;
            const half_t xh = *reinterpret_cast<const half_t *>(rowbuf + m * H * 2 + off * 2);
            if constexpr (QOP >= 2) {
              const half_t wg = *reinterpret_cast<const half_t *>(wbuf + off * 2);
              v[k] = (float)(half_t)(((float)xh * (float)wg) * rv);                    // RMSNorm.cuh:145-151
            } else {
              v[k] = (float)xh;
            }
          }
        }
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[i]));
        amax = max8(amax);
        const GroupScale gs = group_scale<false>(amax, keeper, p.q_clip);
        float tr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) tr[i] = group_code<false>(v[i], gs);
        const v4u w = pack_codes16(tr, keeper);
        if (keeper) *reinterpret_cast<v4u *>(qa8 + m * kKeeper + j * 16) = w;
        else *reinterpret_cast<v2u *>(qa4 + m * K4h + g * 64 + j * 8) = v2u{w[0], w[1]};
        if (j == 0) {
          if (keeper) qsa8[m] = f2h(gs.s_store);
          else qsa[g * MQ + m] = f2h(gs.s_store);
        }
      }
    }
    lds_barrier();
  }
  v4u a[CNT];
  // token scales stay RAW (the 16 loaded bits) until their use: declared as half_t the compiler converts each one to float right
  // behind its load -- `global_load_ushort; global_load_dwordx4; s_waitcnt vmcnt(1); v_cvt_f32_f16` per group -- i.e. it waits for
  // everything issued so far before it issues the next group's loads: CNT dependent round trips instead of one (round 2 shipped
  // that: 0.1-0.2 us of every launch with the operands in L2).  opaque_half() is the use-site conversion the scheduler cannot hoist.
  unsigned sa[CNT];
  const int mq = min(row, p.M - 1);                         // QOP: my token's row in the LDS operand
  auto load_act = [&](int tb, int j) {
    if constexpr (QOP != 0) a[j] = *reinterpret_cast<const v4u *>(qa4 + mq * K4h + (i0 + j) * 64 + kb * 16);
    else a[j] = *reinterpret_cast<const v4u *>(abase + j * 64 + aoff[tb]);
  };
  auto load_sa = [&](int tb, int j) {
    if constexpr (QOP != 0) sa[j] = __builtin_bit_cast(unsigned short, qsa[(i0 + j) * MQ + mq]);
    else sa[j] = *reinterpret_cast<const unsigned short *>(sabase + (int64_t)j * p.ldA * 2 + soff[tb]);
  };
  auto opaque_half = [](unsigned raw) {
    asm volatile("" : "+v"(raw));
    return (float)__builtin_bit_cast(half_t, (unsigned short)raw);
  };
  if constexpr (QOP == 0) {
#pragma unroll
    for (int j = 0; j < CNT; ++j)
      if (j < ng) { load_act(0, j); load_sa(0, j); }
  }
  v4u ak[2] = {};
  unsigned sak = 0;
  auto load_keeper_act = [&](int tb) {
    if constexpr (QOP != 0) {
      const uint8_t *kp = qa8 + mq * kKeeper + kb * 16;
      ak[0] = *reinterpret_cast<const v4u *>(kp);
      ak[1] = *reinterpret_cast<const v4u *>(kp + 64);
      sak = __builtin_bit_cast(unsigned short, qsa8[mq]);
      return;
    }
    const char *kp = reinterpret_cast<const char *>(p.A8) + koff[tb];
    ak[0] = *reinterpret_cast<const v4u *>(kp);
    ak[1] = *reinterpret_cast<const v4u *>(kp + 64);
    sak = *reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(p.sA8) + soff[tb]);
  };
  if constexpr (QOP == 0) {
    if (keeper) load_keeper_act(0);
  }

  for (int blk = blk0; blk < blk1; ++blk, n0 += 16) {       // (one block unless QOP)
  const bool more = LOOP && blk + 1 < blk1;                 // workgroup-uniform: the next block's weights are requested in place
  if constexpr (QOP != 0) {
    // the token rows' packed operand is read out of LDS again for every block: kept in registers across the block loop it would
    // hold 4 CNT + CNT of them for the whole kernel (the 14-slot instance then spills 50 registers)
#pragma unroll
    for (int j = 0; j < CNT; ++j)
      if (j < ng) { load_act(0, j); load_sa(0, j); }
    if (keeper) load_keeper_act(0);
  }
  float (*part)[MBLK][64][4] = reinterpret_cast<float (*)[MBLK][64][4]>(lds_raw + (size_t)(QOP != 0 ? ((blk - blk0) & 1) : 0) * NW * MBLK * 64 * 16);
  float c[MBLK][4];
#pragma unroll
  for (int tb = 0; tb < MBLK; ++tb)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[tb][r] = 0.f;

#pragma unroll
  for (int tb = 0; tb < MBLK; ++tb) {
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
      __builtin_amdgcn_sched_barrier(0);                   // keep the refills where they are written (register pressure)
      if (j < ng) {                                        // wave-uniform
        const v4i be = even_codes(a[j]), bo = odd_codes(a[j]);
        if (tb + 1 < MBLK) load_act(tb + 1, j);            // in place: the next block's chunk is on its way during this block
        const v4i ae = even_codes(w[j]), ao = odd_codes(w[j]);
        if constexpr (LOOP) {
          if (more) {                                      // the next feature block: 16 rows further
            const v4u *wp = reinterpret_cast<const v4u *>(wbase + (int64_t)(blk + 1 - blk0) * 16 * K4h + j * 64 + woff);
            w[j] = NT ? __builtin_nontemporal_load(wp) : *wp;
          }
        }
        v4i acc = {0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(ae, be, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(ao, bo, acc, 0, 0, 0);
        dequant4(acc, opaque_half(sa[j]) * (1.0f / 256.0f), sb[j], c[tb]);
        if constexpr (LOOP) {
          if (more) sb[j] = *reinterpret_cast<const v2u *>(sbbase + (int64_t)(blk + 1 - blk0) * 32 + (int64_t)j * p.N * 2 + 8 * kb);
        }
        if (tb + 1 < MBLK) load_sa(tb + 1, j);
      }
    }
    if (keeper) {                                          // INT8 keeper, last: both 64-column k-steps into ONE accumulator, one
      const float sa8 = opaque_half(sak);                  // de-quantisation (the contract; Dense_layer_gemm_i4_o16.cuh:640-691)
      v4i acc = {0, 0, 0, 0};
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf)
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(v4i, wk[hlf]), __builtin_bit_cast(v4i, ak[hlf]), acc, 0, 0, 0);
      dequant4(acc, sa8, sbk, c[tb]);
      if constexpr (LOOP) {
        if (more) {
          const char *kp = reinterpret_cast<const char *>(p.B8) + (int64_t)(n0 + 16) * kKeeper + (unsigned)(row * kKeeper + kb * 16);
          wk[0] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4u *>(kp)) : *reinterpret_cast<const v4u *>(kp);
          wk[1] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4u *>(kp + 64)) : *reinterpret_cast<const v4u *>(kp + 64);
          sbk = *reinterpret_cast<const v2u *>(reinterpret_cast<const char *>(p.sB8 + n0 + 16) + 8 * kb);
        }
      }
      if (tb + 1 < MBLK) load_keeper_act(tb + 1);
    }
  }

  // ---- partial sums of the NW waves, added in wave order (QOP: two areas in alternation -- wave 0 may still be adding block b's
  // when the other waves write block b + 1's; it has passed this barrier again before anybody writes block b + 2's)
#pragma unroll
  for (int tb = 0; tb < MBLK; ++tb)
    *reinterpret_cast<v4f *>(&part[wave][tb][lane][0]) = v4f{c[tb][0], c[tb][1], c[tb][2], c[tb][3]};
  __syncthreads();
  for (int tb = wave; tb < MBLK; tb += NW) {
    v4f s = *reinterpret_cast<const v4f *>(&part[0][tb][lane][0]);
#pragma unroll
    for (int w2 = 1; w2 < NW; ++w2) {
      const v4f q = *reinterpret_cast<const v4f *>(&part[w2][tb][lane][0]);
      s = v4f{s[0] + q[0], s[1] + q[1], s[2] + q[2], s[3] + q[3]};
    }
    const int m = tb * 16 + row;
    if constexpr (OUT == 2) {
      if (m < p.M) {
        const int seg = n0 / p.seg_n, nl = n0 - seg * p.seg_n + 4 * kb;       // wave-uniform segment (seg_n is a multiple of 16)
        void *out = seg == 0 ? p.seg_out[0] : (seg == 1 ? p.seg_out[1] : p.seg_out[2]);
        if ((p.seg_f32 >> seg) & 1u) {
          *reinterpret_cast<v4f *>(reinterpret_cast<float *>(out) + (int64_t)m * p.seg_n + nl) = s;
        } else {
          v2u o;
          half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = f2h(s[r]);
          if (seg == 0 && p.seg_add) {                   // fp16 + fp16 as torch adds halves: one FP32 addition, one rounding
            const v2u a = *reinterpret_cast<const v2u *>(p.seg_add + (int64_t)m * p.seg_n + nl);
            const half_t *av = reinterpret_cast<const half_t *>(&a);
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = f2h((float)ov[r] + (float)av[r]);
          }
          *reinterpret_cast<v2u *>(reinterpret_cast<half_t *>(out) + (int64_t)m * p.seg_n + nl) = o;
        }
      }
    } else if (m < p.M && OUT32) {
      *reinterpret_cast<v4f *>(p.ws + (int64_t)m * p.N + n0 + 4 * kb) = s;
    } else if (m < p.M) {
      v2u o;
      half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = f2h(s[r]);
      *reinterpret_cast<v2u *>(p.D + (int64_t)m * p.N + n0 + 4 * kb) = o;
    }
  }
  }   // feature blocks of this workgroup
}

template <int NW, int MBLK, int CNT, int OUT = 0, bool NT = false>
static int launch(const GemmParams &p, hipStream_t s) {
  constexpr size_t lds = (size_t)NW * MBLK * 64 * 16;
  if constexpr (lds > 64 * 1024) {
    static std::atomic<uint64_t> attr_done{0};
    if (ensure_max_lds(reinterpret_cast<const void *>(&gemm_w4a4_skinny_kernel<NW, MBLK, CNT, OUT, NT>), (int)lds, attr_done) != ATOM_OK)
      return ATOM_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((gemm_w4a4_skinny_kernel<NW, MBLK, CNT, OUT, NT>), dim3((unsigned)(p.N / 16)), dim3(NW * 64), lds, s, p);
  return check_launch();
}

// one or two tokens with the preceding quantiser inside the launch (p.q_op)
template <int NW, int CNT, int OUT, int QOP>
static int launch_q1(const GemmParams &p, hipStream_t s) {
  static_assert(NW == kQWaves, "skinny_q_fits() is written for the 8-wave workgroup");
  if (!skinny_q_fits(QOP, p.M, 2 * p.K4h + kKeeper)) return ATOM_ERR_SHAPE;
  const size_t lds = q_lds_bytes(QOP, p.K4h, p.G);
  static std::atomic<uint64_t> attr_done{0};
  if (ensure_max_lds(reinterpret_cast<const void *>(&gemm_w4a4_skinny_kernel<NW, 1, CNT, OUT, true, QOP>), kQLdsMax, attr_done) != ATOM_OK)
    return ATOM_ERR_LAUNCH;
  // one workgroup per CU at most, each with the consecutive feature blocks of its share (round 6: one quantiser per workgroup,
  // not per block; N = 4096 keeps its 256 single-block workgroups)
  const int nblk = p.N / 16, cap = CNT <= 8 ? ATOM_TUNE("ATOM_SKINNY_Q_GRID", 256) : nblk;   // (the kernel's LOOP)
  hipLaunchKernelGGL((gemm_w4a4_skinny_kernel<NW, 1, CNT, OUT, true, QOP>), dim3((unsigned)(nblk < cap ? nblk : cap)), dim3(NW * 64), lds, s, p);
  return check_launch();
}
template <int NW, int CNT, int OUT>
static int launch_q(const GemmParams &p, hipStream_t s) {
  switch (p.q_op) {
    case 1: return launch_q1<NW, CNT, OUT, 1>(p, s);
    case 2: return launch_q1<NW, CNT, OUT, 2>(p, s);
    case 3: return launch_q1<NW, CNT, OUT, 3>(p, s);
    case 4: return launch_q1<NW, CNT, OUT, 4>(p, s);
  }
  return ATOM_ERR_INVALID_ARG;
}

template <int NW, int CNT, int OUT = 0>
static int launch_m(const GemmParams &p, hipStream_t s) {
  const int mblk = (p.M + 15) / 16;
  if (mblk <= 1) return p.M <= ATOM_TUNE("ATOM_SKINNY_NT_MAXM", 2) ? launch<NW, 1, CNT, OUT, true>(p, s) : launch<NW, 1, CNT, OUT>(p, s);
  if constexpr (!(NW == 8 && CNT == 14)) {              // (that instance spills 14 VGPRs; the 4-block one does not)
    if (mblk <= 2) return launch<NW, 2, CNT, OUT>(p, s);
  }
  if (mblk <= 4) return launch<NW, 4, CNT, OUT>(p, s);
  if constexpr (CNT <= 8) {
    if (mblk <= 8) return launch<NW, 8, CNT, OUT>(p, s);
    if (mblk <= 16) return launch<NW, 16, CNT, OUT>(p, s);
  }
  return ATOM_ERR_SHAPE;
}

// u4 epilogue of the decode path (reference: DenseLayerGEMM_i4_o4.cu:704-788; same arithmetic as the tile kernel's epilogue
// in gemm_w4a4_v2.hip): every 128-column group of a row of the FP32 sums -> scale = (max-min)/15, zero = -min,
// q = clamp(round_half_away((x + zero) * (1/scale)), 0, 15).  Half a wave per group, 4 values per lane.
__global__ __launch_bounds__(256) void o4_quant_kernel(const float *__restrict__ x, uint8_t *__restrict__ q, half_t *__restrict__ sz,
                                                        int64_t groups, int gpr /* groups per row */, int ref_extrema) {
  const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
  const int l = threadIdx.x & 31;
  if (grp >= groups) return;
  const int64_t m = grp / gpr;
  const int g = (int)(grp % gpr);
  const v4f v = *reinterpret_cast<const v4f *>(x + (m * gpr + g) * 128 + 4 * l);
  const v4f e = ref_extrema ? v4f{fabsf(v[0]), fabsf(v[1]), fabsf(v[2]), fabsf(v[3])} : v;   // ATOM_O4_REF_EXTREMA: extrema of |x|
  float lo = fminf(fminf(e[0], e[1]), fminf(e[2], e[3])), hi = fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3]));
#pragma unroll
  for (int k = 16; k >= 1; k >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, k));
    hi = fmaxf(hi, __shfl_xor(hi, k));
  }
  const float scale = (hi - lo) / 15.f, zero = -lo, rs = 1.0f / scale;
  unsigned w = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) w |= (ref_extrema ? o4_code<true>(v[k], zero, rs, scale) : o4_code<false>(v[k], zero, rs, scale)) << (4 * k);
  *reinterpret_cast<unsigned short *>(q + (m * gpr + g) * 64 + 2 * l) = (unsigned short)w;
  if (l == 0) {
    half_t *d = sz + (m * gpr + g) * 2;
    d[0] = f2h(scale);
    d[1] = f2h(zero);
  }
}

}  // namespace skinny

// 1 <= M <= 256 (<= 64 when K > 8192), reference packed format, K <= 14336.  ATOM_ERR_SHAPE when K is too long for the register-resident weight slice
// (caller falls back to the tile kernels).
int launch_gemm_skinny(const GemmParams &p, hipStream_t s) {
  if (p.M > 256 || p.a_wide || p.f6_rows_a || (p.N % 16) != 0) return ATOM_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(p.D) & 7u) != 0 || (reinterpret_cast<uintptr_t>(p.sB) & 7u) != 0 ||
      (reinterpret_cast<uintptr_t>(p.sB8) & 7u) != 0)
    return ATOM_ERR_SHAPE;                                 // 8-byte scale loads / stores: the tile kernels take these
  const int per = (p.G + 1 + 7) / 8;                      // 8 waves split K (measured: 4 waves with twice the slice are slower)
  if (per > skinny::CNT_MAX) return ATOM_ERR_SHAPE;
  if (per <= 4) return skinny::launch_m<8, 4>(p, s);
  return per <= 8 ? skinny::launch_m<8, 8>(p, s) : skinny::launch_m<8, 14>(p, s);
}

// FP32 sums [M, N] into p.ws (no final rounding): the k / v projections of a decode step, ahead of the u4 epilogue.  Same
// shapes as launch_gemm_skinny.
int launch_gemm_skinny_f32(const GemmParams &p, hipStream_t s) {
  if (p.M > 256 || p.a_wide || p.f6_rows_a || (p.N % 16) != 0 || !p.ws) return ATOM_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(p.sB) & 7u) != 0 || (reinterpret_cast<uintptr_t>(p.sB8) & 7u) != 0) return ATOM_ERR_SHAPE;
  const int per = (p.G + 1 + 7) / 8;
  if (per > skinny::CNT_MAX) return ATOM_ERR_SHAPE;
  return per <= 4 ? skinny::launch_m<8, 4, 1>(p, s)
                  : (per <= 8 ? skinny::launch_m<8, 8, 1>(p, s) : skinny::launch_m<8, 14, 1>(p, s));
}

// Segmented outputs (atom_gemm_w4a4_multi): one launch for the projections that share an activation operand -- q / k / v, gate / up
// -- or for one projection + the residual add.  Per-feature arithmetic and summation order are those of launch_gemm_skinny /
// launch_gemm_skinny_f32 (a workgroup owns 16 features of ONE segment): bit-identical to the separate launches.
// ... with the quantiser that precedes the GEMM inside the launch (p.q_op; 1 or 2 tokens)
// THE shape predicate of the quantiser-in-front launch, shared by atom_gemm_w4a4_multi_q_fits and the launcher (round 3 had two, and the
// query was looser than the launch for ops 1-3): per thread of the 512 at most 2 slot tasks of 16 channels (SiLU x up: 3), at most 3
// 16-byte chunks of the token rows and 2 of the norm weight (ops 1-3), and everything within 96 KiB of LDS.
bool skinny_q_fits(int q_op, int64_t M, int64_t H) {
  if (q_op < 1 || q_op > 4 || M < 1 || M > 2 || H < 2 * kKeeper || ((H - kKeeper) % kGroup) != 0) return false;
  const int64_t threads = skinny::kQWaves * 64;
  if (M * (H >> 4) > (q_op == 4 ? 3 : 2) * threads) return false;
  if (q_op <= 3 && (M * (H >> 3) > 3 * threads || (H >> 3) > 2 * threads)) return false;
  const int K4h = (int)((H - kKeeper) / 2), G = (int)((H - kKeeper) / kGroup);
  return skinny::q_lds_bytes(q_op, K4h, G) <= (size_t)skinny::kQLdsMax;
}

int launch_gemm_skinny_multi_q(const GemmParams &p, hipStream_t s) {
  if (p.M > 2 || p.q_op < 1 || p.q_op > 4 || (p.N % 16) != 0 || p.seg_n < 16 || (p.seg_n % 16) != 0 || (p.N % p.seg_n) != 0 || p.N / p.seg_n > 3)
    return ATOM_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(p.sB) & 7u) != 0 || (reinterpret_cast<uintptr_t>(p.sB8) & 7u) != 0) return ATOM_ERR_SHAPE;
  const int per = (p.G + 1 + 7) / 8;
  if (per > skinny::CNT_MAX) return ATOM_ERR_SHAPE;
  return per <= 4 ? skinny::launch_q<8, 4, 2>(p, s) : (per <= 8 ? skinny::launch_q<8, 8, 2>(p, s) : skinny::launch_q<8, 14, 2>(p, s));
}

int launch_gemm_skinny_multi(const GemmParams &p, hipStream_t s) {
  if (p.M > 256 || p.a_wide || p.f6_rows_a || (p.N % 16) != 0 || p.seg_n < 16 || (p.seg_n % 16) != 0 || (p.N % p.seg_n) != 0 ||
      p.N / p.seg_n > 3)
    return ATOM_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(p.sB) & 7u) != 0 || (reinterpret_cast<uintptr_t>(p.sB8) & 7u) != 0) return ATOM_ERR_SHAPE;
  const int per = (p.G + 1 + 7) / 8;
  if (per > skinny::CNT_MAX) return ATOM_ERR_SHAPE;
  return per <= 4 ? skinny::launch_m<8, 4, 2>(p, s)
                  : (per <= 8 ? skinny::launch_m<8, 8, 2>(p, s) : skinny::launch_m<8, 14, 2>(p, s));
}

// Decode path of the u4-output GEMM: the FP32 sums into the caller's workspace, then the u4 epilogue as a second launch
int launch_gemm_skinny_o4(const GemmParams &p, hipStream_t s) {
  if ((p.N % 128) != 0 || !p.D4 || !p.Dsz) return ATOM_ERR_SHAPE;
  const int st = launch_gemm_skinny_f32(p, s);
  if (st != ATOM_OK) return st;
  const int gpr = p.N / 128;
  const int64_t groups = (int64_t)p.M * gpr;
  hipLaunchKernelGGL(skinny::o4_quant_kernel, dim3((unsigned)((groups * 32 + 255) / 256)), dim3(256), 0, s, p.ws, p.D4, p.Dsz,
                     groups, gpr, p.o4_ref);
  return check_launch();
}

}  // namespace atom
