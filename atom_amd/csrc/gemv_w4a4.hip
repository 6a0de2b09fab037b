// W4A4 GEMM, decode path (M <= 7; the kernel itself handles up to 16 rows) for gfx950: weight streaming, HBM-bound.
//
// BASELINE config 2 (M=1, N=K=4096): 8.9 MB of packed weights + scales must cross HBM once; the 256x256 MFMA tile
// kernel would launch 16 workgroups on a 256-CU chip (55 us measured).  Here instead:
//   * one wave owns one output feature n at a time and streams that weight row with fully coalesced 16-byte loads
//     (lane l reads chunk l, l+64, ...: 1 KiB per wave instruction, straight to VGPRs -- a row is read exactly once,
//     so an LDS round trip would be pure overhead);
//   * int4 x int4 dot products run on v_dot8_i32_i4 directly on the PACKED dwords (no widening at all), the INT8
//     keeper on v_dot4_i32_i8;
//   * the (at most 16) activation rows live in LDS (34 KB at K=4096), read with conflict-free ds_read_b128;
//   * a quantisation group is 4 consecutive chunks = 4 consecutive lanes: integer partials are combined exactly with
//     two DPP xor-adds, the quad leader applies  c = fma(idot, sA[m,g]*sB[g,n], c) (the scale product is exact);  per-lane sums
//     are then reduced with a 6-step butterfly, keeper added last (FP32 summation ORDER therefore differs from the
//     prefill kernel; both are within 1 fp16 ulp of the exact value).
//   * grid: up to 1024 waves (4 per workgroup) round-robin over the N rows.
// Replaces the M<=16 rows of the reference's NVBench sweep (kernels/src/GEMM/bench_dense_layer_gemm_i4_o16.cu:64-69),
// which runs the same 128x128 tensor-core tile kernel for every M.
#include "common.h"

namespace atom {

// xor-1 / xor-2 lane exchange inside a quad as a DPP modifier (no LDS-pipe ds_bpermute)
__device__ __forceinline__ int quad_sum(int d) {
  d += __builtin_amdgcn_mov_dpp(d, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  d += __builtin_amdgcn_mov_dpp(d, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  return d;
}

template <int MB>   // rows of activations handled per pass (compile-time: 1, 2, 4, 8, 16)
__global__ __launch_bounds__(256) void gemv_w4a4_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K4h = p.K4h, G = p.G;
  const int rowb = K4h + kKeeper;                       // bytes of one activation row in LDS: packed int4 | int8 keeper
  float *sAl = reinterpret_cast<float *>(lds + ((MB * rowb + 15) & ~15));   // [MB][G+1] activation scales (keeper last)

  // stage activations (tiny) into LDS
  for (int m = 0; m < MB; ++m) {
    const int ms = min(m, p.M - 1);
    for (int i = tid * 16; i < K4h; i += 256 * 16)
      *reinterpret_cast<v4u *>(lds + m * rowb + i) = *reinterpret_cast<const v4u *>(p.A4 + (int64_t)ms * K4h + i);
    if (tid < 8)
      *reinterpret_cast<v4u *>(lds + m * rowb + K4h + tid * 16) =
          *reinterpret_cast<const v4u *>(p.A8 + (int64_t)ms * kKeeper + tid * 16);
    const int off = p.ref_layout ? ref_scale_index(ms) : ms;
    for (int g = tid; g <= G; g += 256)
      sAl[m * (G + 1) + g] = g < G ? (float)p.sA[(int64_t)g * p.ldA + off] : (float)p.sA8[off];
  }
  __syncthreads();

  const int nchunks = K4h >> 4;                         // 16-byte chunks per weight row (multiple of 4)
  const int nwaves = gridDim.x * 4;
  for (int n = blockIdx.x * 4 + wave; n < p.N; n += nwaves) {
    const uint8_t *brow = p.B4 + (int64_t)n * K4h;
    float acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = 0.f;

    for (int c0 = 0; c0 < nchunks; c0 += 64) {
      const int c = c0 + lane;
      const bool ok = c < nchunks;
      v4i w = {0, 0, 0, 0};
      float sb = 0.f;
      const int g = c >> 2;
      if (ok) {
        w = *reinterpret_cast<const v4i *>(brow + c * 16);
        if ((lane & 3) == 0) sb = (float)p.sB[(int64_t)g * p.N + n];
      }
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        int d = 0;
        if (ok) {
          const v4i a = *reinterpret_cast<const v4i *>(lds + m * rowb + c * 16);
          d = __builtin_amdgcn_sdot8(a[0], w[0], d, false);
          d = __builtin_amdgcn_sdot8(a[1], w[1], d, false);
          d = __builtin_amdgcn_sdot8(a[2], w[2], d, false);
          d = __builtin_amdgcn_sdot8(a[3], w[3], d, false);
        }
        d = quad_sum(d);                                // exact: the group's 128-element integer dot
        if (ok && (lane & 3) == 0) {
          acc[m] = __builtin_fmaf((float)d, sAl[m * (G + 1) + g] * sb, acc[m]);   // the contract: exact scale product, one rounding
        }
      }
    }
    // INT8 keeper: 8 chunks of 16 bytes on lanes 0..7
    {
      v4i w = {0, 0, 0, 0};
      if (lane < 8) w = *reinterpret_cast<const v4i *>(p.B8 + (int64_t)n * kKeeper + lane * 16);
      const float sb8 = (float)p.sB8[n];
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        int d = 0;
        if (lane < 8) {
          const v4i a = *reinterpret_cast<const v4i *>(lds + m * rowb + K4h + lane * 16);
          d = __builtin_amdgcn_sdot4(a[0], w[0], d, false);
          d = __builtin_amdgcn_sdot4(a[1], w[1], d, false);
          d = __builtin_amdgcn_sdot4(a[2], w[2], d, false);
          d = __builtin_amdgcn_sdot4(a[3], w[3], d, false);
        }
        d = quad_sum(d);
        d += __shfl_xor(d, 4);
        float s = acc[m];
        s = wave_sum_butterfly(s);                      // xor 32, 16, .., 1 without the LDS pipeline (common.h)
        if (lane == 0 && m < p.M) {
          p.D[(int64_t)m * p.N + n] = f2h(__builtin_fmaf((float)d, sAl[m * (G + 1) + G] * sb8, s));
        }
      }
    }
  }
}

// M = 1 (BASELINE config 2): no LDS, no barrier -- every lane asks for its weight chunk (HBM), the matching activation chunk and the
// group's two scales (L2) in the same few instructions, so a K batch is ONE memory round trip deep.
// Round 4 (tools/probes/gemv_probe.cpp, profiles/r04/gemv_probe.txt): the round-3 form of this kernel kept a whole row (up to 4 chunks
// per lane, their activations and scales) in registers -- 65-107 VGPRs, i.e. 4-7 waves per SIMD -- and reached 0.23-0.33 of 8 TB/s cold
// where a plain streaming read of the same bytes reaches 0.39 (8.9 MB) to 0.69 (37 MB).  What the probe showed:
//   * the read rate is set by how many waves are resident, not by how much one wave has in flight: ONE 16-byte chunk per feature and
//     batch (22-29 VGPRs, every wave slot of the CU filled) beats 2-4 chunks per lane on every shape;
//   * workgroup b runs on XCD b % 8 and a 64-byte line of weight scales holds 32 adjacent features of one group: with the features
//     dealt round-robin every XCD's L2 fetched every line (8 x the scale traffic, +23 % bytes at 4096 x 4096); each XCD now owns a
//     contiguous eighth of the features;
//   * two adjacent features per wave (R = 2) share the activation chunk and the token scale: better from 8192 features up;
//   * staging the activation row and the scale tile in LDS once per workgroup measured equal to reading them per wave from L2.
// 1 x 13824 x 5120: 16.6 -> 10.2 us cold (0.28 -> 0.47 of 8 TB/s), 1 x 4096 x 4096: 4.86 -> 4.13 (0.23 -> 0.27; the plain read of
// these bytes takes 2.8 us, 1.6 of them the launch).  Same per-lane arithmetic and summation order as before: lane l owns chunks
// l, l + 64, ... in ascending order, a quad sums a group exactly, the quad leader applies the two scales (bit-identical output).
// One or two tokens (MT rows of activations; round 4): the weight chunk of a feature is loaded once and meets every token's
// activation chunk (L2) -- per token 4 v_dot8 + the quad sum + 3 VALU per chunk.  Two tokens: 2 x 5120 x 13824 cold 12.5 us where the
// MFMA decode-batch kernel takes 17.5; from three tokens the VALU work loses to it (MT = 4, 8: tuning builds; profiles/r04/decode_small_m.txt).
// Every token's sum is formed exactly as the one-token kernel forms it (same lanes, same order): M rows = M independent one-token results.
// OUT 0: fp16 D [M, N]; 1: the FP32 sums into p.ws [M, N] (atom_gemm_w4a4_f32); 2: segmented outputs (atom_gemm_w4a4_multi: p.seg_*)
template <int R, int MT, int OUT>
__global__ __launch_bounds__(256) void gemv1_w4a4_kernel(GemmParams p) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {  // the kernel arguments in ONE batch of scalar loads (hipcc fetched them in three, a scalar-cache round trip apart, in front of the
     // first request of a kernel that is one trip to HBM long)
    const void *a0 = p.A4, *a1 = p.B4, *a2 = p.sA, *a3 = p.sB, *a4 = p.A8, *a5 = p.B8, *a6 = p.sA8, *a7 = p.sB8, *a8_ = p.D;
    const int i0 = p.M, i1 = p.N, i2 = p.K4h, i3 = p.G, i4 = p.ref_layout, i5 = (int)gridDim.x;   // (gridDim: an implicit argument behind the struct)
    asm volatile("" ::"s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(a6), "s"(a7), "s"(a8_), "s"(i0), "s"(i1), "s"(i2), "s"(i3), "s"(i4), "s"(i5));
  }
  const int K4h = p.K4h;
  const int nchunks = K4h >> 4;
  const bool leader = (lane & 3) == 0;
  // bijective XCD-aware map: XCD x = b % 8 takes the blocks [x * q + min(x, rem), ...) of the feature axis
  const int q = (int)gridDim.x >> 3, rem = (int)gridDim.x & 7;
  const int x = blockIdx.x & 7;
  const int blk = x * q + min(x, rem) + (blockIdx.x >> 3);
  const int n0 = (blk * 4 + wave) * R;
  if (n0 >= p.N) return;                                  // (wave-uniform)
  int nr[R], mr[MT], so[MT];                              // clamped feature / token indices, token offsets inside a scale row
#pragma unroll
  for (int r = 0; r < R; ++r) nr[r] = min(n0 + r, p.N - 1);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    mr[m] = min(m, p.M - 1);
    so[m] = p.ref_layout ? ref_scale_index(mr[m]) : mr[m];
  }
  float acc[R][MT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[r][m] = 0.f;
  const unsigned short *sBu = reinterpret_cast<const unsigned short *>(p.sB), *sAu = reinterpret_cast<const unsigned short *>(p.sA);
  // the keeper's operands are requested with the first instructions and consumed after the K loop (behind the loop they would be one
  // more dependent round trip at the very end of every wave)
  v4i a8[MT], w8[R];
  unsigned short sa8u[MT], sb8u[R];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    a8[m] = *reinterpret_cast<const v4i *>(p.A8 + (int64_t)mr[m] * kKeeper + (lane & 7) * 16);   // lanes 8.. hold copies; lane 0's sums touch lanes 0-7 only
    sa8u[m] = reinterpret_cast<const unsigned short *>(p.sA8)[so[m]];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    w8[r] = *reinterpret_cast<const v4i *>(p.B8 + (int64_t)nr[r] * kKeeper + (lane & 7) * 16);
    sb8u[r] = reinterpret_cast<const unsigned short *>(p.sB8)[nr[r]];
  }
  // Round 6: rows of at most two batches (K_total <= 4224: BASELINE config 2) request BOTH weight chunks -- the
  // only HBM traffic of the wave -- before anything else, then both batches' activation chunks and scales (L2), and compute after:
  // the run-time loop below issues batch 1's requests behind batch 0's arithmetic, i.e. two dependent trips to HBM per wave where
  // the launch itself is 1.6 us of a 4 us kernel.  Same per-lane order (batch 0, then batch 1): same bits.
  if (nchunks <= 128) {
    const int cc0 = min(lane, nchunks - 1), cc1 = min(64 + lane, nchunks - 1);
    v4i w0[R], w1[R], a0[MT], a1[MT];
    unsigned short sb0[R], sb1[R], sa0[MT], sa1[MT];
#pragma unroll
    for (int r = 0; r < R; ++r) w0[r] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p.B4 + (int64_t)nr[r] * K4h + cc0 * 16));
#pragma unroll
    for (int r = 0; r < R; ++r) w1[r] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p.B4 + (int64_t)nr[r] * K4h + cc1 * 16));
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      a0[m] = *reinterpret_cast<const v4i *>(p.A4 + (int64_t)mr[m] * K4h + cc0 * 16);
      a1[m] = *reinterpret_cast<const v4i *>(p.A4 + (int64_t)mr[m] * K4h + cc1 * 16);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      sb0[r] = sBu[(int64_t)(cc0 >> 2) * p.N + nr[r]];
      sb1[r] = sBu[(int64_t)(cc1 >> 2) * p.N + nr[r]];
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      sa0[m] = sAu[(int64_t)(cc0 >> 2) * p.ldA + so[m]];
      sa1[m] = sAu[(int64_t)(cc1 >> 2) * p.ldA + so[m]];
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const bool ok = b * 64 + lane < nchunks;
      if (b == 1 && nchunks <= 64) break;                   // (wave-uniform: a single batch)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float saf = (float)__builtin_bit_cast(half_t, b ? sa1[m] : sa0[m]);
        const v4i &am = b ? a1[m] : a0[m];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const v4i &wr = b ? w1[r] : w0[r];
          int d = 0;
          d = __builtin_amdgcn_sdot8(am[0], wr[0], d, false);
          d = __builtin_amdgcn_sdot8(am[1], wr[1], d, false);
          d = __builtin_amdgcn_sdot8(am[2], wr[2], d, false);
          d = __builtin_amdgcn_sdot8(am[3], wr[3], d, false);
          d = quad_sum(d);
          const float next = __builtin_fmaf((float)d, saf * (float)__builtin_bit_cast(half_t, b ? sb1[r] : sb0[r]), acc[r][m]);
          acc[r][m] = (leader && ok) ? next : acc[r][m];
        }
      }
    }
  } else
  for (int c0 = 0; c0 < nchunks; c0 += 64) {
    const int c = c0 + lane;
    const bool ok = c < nchunks;
    // NO divergent control flow around the loads: a lane past the end re-reads the last chunk and every lane of a quad reads the
    // group's scales (same address: one request).  Inside `if (ok) / if (leader)` blocks the compiler waits for each 16-bit scale
    // load before leaving the block -- and, vmcnt being in order, for the weight chunk issued before it: two or three dependent
    // round trips per batch instead of one (the round-4 rewrite measured 4.7 us instead of 4.1 at 4096 x 4096 until this was gone).
    const int cc = min(c, nchunks - 1);
    v4i w[R], a[MT];
    unsigned short sbu[R], sau[MT];
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p.B4 + (int64_t)nr[r] * K4h + cc * 16));   // nt: read once, by this CU only
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const v4i *>(p.A4 + (int64_t)mr[m] * K4h + cc * 16);
#pragma unroll
    for (int r = 0; r < R; ++r) sbu[r] = sBu[(int64_t)(cc >> 2) * p.N + nr[r]];
#pragma unroll
    for (int m = 0; m < MT; ++m) sau[m] = sAu[(int64_t)(cc >> 2) * p.ldA + so[m]];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float saf = (float)__builtin_bit_cast(half_t, sau[m]);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int d = 0;
        d = __builtin_amdgcn_sdot8(a[m][0], w[r][0], d, false);
        d = __builtin_amdgcn_sdot8(a[m][1], w[r][1], d, false);
        d = __builtin_amdgcn_sdot8(a[m][2], w[r][2], d, false);
        d = __builtin_amdgcn_sdot8(a[m][3], w[r][3], d, false);
        d = quad_sum(d);                                    // exact: the group's 128-element integer dot
        const float next = __builtin_fmaf((float)d, saf * (float)__builtin_bit_cast(half_t, sbu[r]), acc[r][m]);   // exact scale product
        acc[r][m] = (leader && ok) ? next : acc[r][m];
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float sa8f = (float)__builtin_bit_cast(half_t, sa8u[m]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int d = 0;
      d = __builtin_amdgcn_sdot4(a8[m][0], w8[r][0], d, false);
      d = __builtin_amdgcn_sdot4(a8[m][1], w8[r][1], d, false);
      d = __builtin_amdgcn_sdot4(a8[m][2], w8[r][2], d, false);
      d = __builtin_amdgcn_sdot4(a8[m][3], w8[r][3], d, false);
      d = quad_sum(d);
      d += __builtin_amdgcn_update_dpp(0, d, 0x104, 0xF, 0xF, true);   // row_shl:4 -- lane 0 += lane 4 (the other lanes' values are not used)
      float s = acc[r][m];
      s = wave_sum_butterfly(s);                      // xor 32, 16, .., 1 without the LDS pipeline (common.h)
      if (lane == 0 && n0 + r < p.N && m < p.M) {
        const float c = __builtin_fmaf((float)d, sa8f * (float)__builtin_bit_cast(half_t, sb8u[r]), s);
        const int n = n0 + r;
        if constexpr (OUT == 0) {
          p.D[(int64_t)m * p.N + n] = f2h(c);
        } else if constexpr (OUT == 1) {
          p.ws[(int64_t)m * p.N + n] = c;
        } else {
          const int seg = n / p.seg_n, nl = n - seg * p.seg_n;
          void *out = seg == 0 ? p.seg_out[0] : (seg == 1 ? p.seg_out[1] : p.seg_out[2]);
          const int64_t at = (int64_t)m * p.seg_n + nl;
          if ((p.seg_f32 >> seg) & 1u) {
            reinterpret_cast<float *>(out)[at] = c;
          } else {
            half_t h = f2h(c);
            if (seg == 0 && p.seg_add) h = f2h((float)h + (float)p.seg_add[at]);   // fp16 + fp16 as torch adds halves (the skinny kernel's rule)
            reinterpret_cast<half_t *>(out)[at] = h;
          }
        }
      }
    }
  }
}

// the few-token dot-product kernel takes M <= kGemvMaxTokens (every entry point: atom_gemm_w4a4_f16 / _f32 / _multi route here)
template <int OUT, int MT>
static int launch_gemv1_mt(const GemmParams &p, hipStream_t s) {
  // one wave per R adjacent output features, every wave resident at once (no feature loop); R = 2 from 8192 features up
  const int R = p.N >= 8192 ? 2 : 1;
  const int blocks = (p.N + 4 * R - 1) / (4 * R);
  if (R == 2) hipLaunchKernelGGL((gemv1_w4a4_kernel<2, MT, OUT>), dim3((unsigned)blocks), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemv1_w4a4_kernel<1, MT, OUT>), dim3((unsigned)blocks), dim3(256), 0, s, p);
  return check_launch();
}
template <int OUT>
static int launch_gemv1_out(const GemmParams &p, hipStream_t s) {
  if (p.M <= 1) return launch_gemv1_mt<OUT, 1>(p, s);
  if (p.M <= 2) return launch_gemv1_mt<OUT, 2>(p, s);
#ifdef ATOM_TOOLS   // (measured slower than the MFMA decode-batch kernel from three tokens up: tuning builds only)
  if (p.M <= 4) return launch_gemv1_mt<OUT, 4>(p, s);
  if (p.M <= 8) return launch_gemv1_mt<OUT, 8>(p, s);
#endif
  return ATOM_ERR_SHAPE;
}
int launch_gemv1(const GemmParams &p, hipStream_t s) { return launch_gemv1_out<0>(p, s); }
// ... the FP32 sums into p.ws / the segmented outputs of atom_gemm_w4a4_multi: the same kernel, the same summation order
int launch_gemv1_f32(const GemmParams &p, hipStream_t s) { return p.ws ? launch_gemv1_out<1>(p, s) : ATOM_ERR_SHAPE; }
int launch_gemv1_multi(const GemmParams &p, hipStream_t s) {
  if (p.seg_n < 1 || (p.N % p.seg_n) != 0 || p.N / p.seg_n > 3 || !p.seg_out[0]) return ATOM_ERR_SHAPE;
  return launch_gemv1_out<2>(p, s);
}

template <int MB>
static int launch_gemv_mb(const GemmParams &p, hipStream_t s) {
  const size_t lds = (size_t)((MB * (p.K4h + kKeeper) + 15) & ~15) + (size_t)MB * (p.G + 1) * 4;
  if (lds > 160 * 1024) return ATOM_ERR_SHAPE;
  static std::atomic<uint64_t> attr_done{0};
  if (ensure_max_lds(reinterpret_cast<const void *>(&gemv_w4a4_kernel<MB>), 160 * 1024, attr_done) != ATOM_OK) return ATOM_ERR_LAUNCH;
  int blocks = (p.N + 3) / 4;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL((gemv_w4a4_kernel<MB>), dim3((unsigned)blocks), dim3(256), lds, s, p);
  return check_launch();
}

// M <= 16.  Returns ATOM_ERR_SHAPE when the activations do not fit LDS (caller falls back to the tile kernel).
int launch_gemv(const GemmParams &p, hipStream_t s) {
  if (p.M <= 1) return launch_gemv1(p, s);
  if (p.M <= 2) return launch_gemv_mb<2>(p, s);
  if (p.M <= 4) return launch_gemv_mb<4>(p, s);
  if (p.M <= 8) return launch_gemv_mb<8>(p, s);
  return launch_gemv_mb<16>(p, s);
}

}  // namespace atom
