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
//     two DPP xor-adds, the quad leader applies  t = round_f32(idot*sA[m,g]); c = fma(t, sB[g,n], c);  per-lane sums
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
          const float t = (float)d * sAl[m * (G + 1) + g];
          acc[m] = __builtin_fmaf(t, sb, acc[m]);
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
          const float t = (float)d * sAl[m * (G + 1) + G];
          p.D[(int64_t)m * p.N + n] = f2h(__builtin_fmaf(t, sb8, s));
        }
      }
    }
  }
}

// M = 1 (BASELINE config 2), second version: no LDS, no barrier.  The staged version above is a chain of three dependent round trips
// (activation row -> LDS, barrier, then the weight row, its scales last); with one token every lane can ask for everything it will
// ever need in its first instructions -- its weight chunks (HBM), the matching activation chunks and the group's two scales (L2:
// every wave reads the same 2 KB row) -- so the launch is ONE memory round trip deep.  Same per-lane arithmetic and summation
// order as gemv_w4a4_kernel<1> (bit-identical output).  UNR chunks per lane are in flight per batch (K = 4096: one batch).
template <int UNR, int ROWS>
__global__ __launch_bounds__(256) void gemv1_w4a4_kernel(GemmParams p) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K4h = p.K4h;
  const int nchunks = K4h >> 4;
  const bool leader = (lane & 3) == 0;
  const int nwaves = gridDim.x * 4;
  // ROWS output features per wave are in flight together (rows n, n + nwaves, ...): with more rows than resident waves the second
  // row's round trip would otherwise start only when the first one's result is stored
  for (int n0 = blockIdx.x * 4 + wave; n0 < p.N; n0 += nwaves * ROWS) {
    v4i w8[ROWS], a8 = {0, 0, 0, 0};
    half_t sb8h[ROWS];
    if (lane < 8) a8 = *reinterpret_cast<const v4i *>(p.A8 + lane * 16);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int n = min(n0 + r * nwaves, p.N - 1);
      w8[r] = v4i{0, 0, 0, 0};
      if (lane < 8) w8[r] = *reinterpret_cast<const v4i *>(p.B8 + (int64_t)n * kKeeper + lane * 16);
      sb8h[r] = p.sB8[n];
    }
    const half_t sa8h = p.sA8[0];
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
    for (int c0 = 0; c0 < nchunks; c0 += 64 * UNR) {
      v4i w[ROWS][UNR], a[UNR];
      half_t sbh[ROWS][UNR], sah[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int c = c0 + u * 64 + lane;
        const bool ok = c < nchunks;
        a[u] = v4i{0, 0, 0, 0};
        sah[u] = (half_t)0;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const int n = min(n0 + r * nwaves, p.N - 1);
          w[r][u] = v4i{0, 0, 0, 0};
          sbh[r][u] = (half_t)0;
          if (ok) {
            w[r][u] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p.B4 + (int64_t)n * K4h + c * 16));   // nt: read once, by this CU only (gemm_w4a4_skinny.hip, NT)
            if (leader) sbh[r][u] = p.sB[(int64_t)(c >> 2) * p.N + n];
          }
        }
        if (ok) {
          a[u] = *reinterpret_cast<const v4i *>(p.A4 + c * 16);
          if (leader) sah[u] = p.sA[(int64_t)(c >> 2) * p.ldA];            // token 0: index 0 in either scale layout
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          int d = 0;
          d = __builtin_amdgcn_sdot8(a[u][0], w[r][u][0], d, false);
          d = __builtin_amdgcn_sdot8(a[u][1], w[r][u][1], d, false);
          d = __builtin_amdgcn_sdot8(a[u][2], w[r][u][2], d, false);
          d = __builtin_amdgcn_sdot8(a[u][3], w[r][u][3], d, false);
          d = quad_sum(d);                                  // exact: the group's 128-element integer dot
          if (leader && c0 + u * 64 + lane < nchunks) {
            const float t = (float)d * (float)sah[u];
            acc[r] = __builtin_fmaf(t, (float)sbh[r][u], acc[r]);
          }
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int n = n0 + r * nwaves;
      int d = 0;
      d = __builtin_amdgcn_sdot4(a8[0], w8[r][0], d, false);
      d = __builtin_amdgcn_sdot4(a8[1], w8[r][1], d, false);
      d = __builtin_amdgcn_sdot4(a8[2], w8[r][2], d, false);
      d = __builtin_amdgcn_sdot4(a8[3], w8[r][3], d, false);
      d = quad_sum(d);
      d += __shfl_xor(d, 4);
      float s = acc[r];
      s = wave_sum_butterfly(s);                      // xor 32, 16, .., 1 without the LDS pipeline (common.h)
      if (lane == 0 && n < p.N) {
        const float t = (float)d * (float)sa8h;
        p.D[n] = f2h(__builtin_fmaf(t, (float)sb8h[r], s));
      }
    }
  }
}

static int launch_gemv1(const GemmParams &p, hipStream_t s) {
  // one wave per output feature, at most 2048 workgroups (the resident set: 8 per CU); beyond that a wave walks its features one
  // after the other.  (Measured, profiles/r02_decode.txt: more workgroups, or two features in flight per wave (ROWS = 2, tuning
  // only), are not faster.)
  const int nb = (p.N + 3) / 4;
  const int rows2 = ATOM_TUNE("ATOM_GEMV1_ROWS2", 0);
  int blocks = rows2 ? (nb + 1) / 2 : nb;
  if (blocks > 2048) blocks = 2048;
  const bool k2 = p.K4h <= 2 * 64 * 16;
  if (rows2) {
    if (k2) hipLaunchKernelGGL((gemv1_w4a4_kernel<2, 2>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemv1_w4a4_kernel<4, 2>), dim3((unsigned)blocks), dim3(256), 0, s, p);
  } else {
    if (k2) hipLaunchKernelGGL((gemv1_w4a4_kernel<2, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemv1_w4a4_kernel<4, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p);
  }
  return check_launch();
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
  if (p.M <= 1) return ATOM_TUNE("ATOM_GEMV1_STAGED", 0) ? launch_gemv_mb<1>(p, s) : launch_gemv1(p, s);
  if (p.M <= 2) return launch_gemv_mb<2>(p, s);
  if (p.M <= 4) return launch_gemv_mb<4>(p, s);
  if (p.M <= 8) return launch_gemv_mb<8>(p, s);
  return launch_gemv_mb<16>(p, s);
}

}  // namespace atom
