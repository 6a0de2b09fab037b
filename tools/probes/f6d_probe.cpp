// Probe (round 4): a weight-streaming decode-batch GEMM on the BF6 MFMA -- the structure of csrc/gemm_w4a4_skinny.hip (one workgroup
// per 16 output features, its 8 waves split the K groups, every wave's weight slice in flight from its first instruction, token blocks
// of 16 re-use the weight registers) with BOTH operands in the F6 format (include/atom_hip.h ATOM_AB_F6 / ATOM_B_F6S), so that a
// (token block, group) costs ONE v_mfma_f32_16x16x128_f8f6f4 + 8 FP32 VALU instead of two INT8 MFMAs + ~45 VALU (widening both
// operands, conversions).  Random operand bytes; timing only (hipGraph replay, hot = one weight set, cold = sets >= 600 MB cycled).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/f6d_probe.cpp -o build/tools/f6d_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef _Float16 half_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
constexpr int PITCH = 104;

struct P {
  const uint8_t *A6, *B6; const float *sB32; const int8_t *A8, *B8; const half_t *sA8, *sB8; half_t *D;
  int M, N, G, Mpad, Npad;
};

__device__ __forceinline__ v8i load_frag(const uint8_t *rec) {       // 24 bytes: this lane's 32 six-bit codes
  const v2u x = *reinterpret_cast<const v2u *>(rec), y = *reinterpret_cast<const v2u *>(rec + 8), z = *reinterpret_cast<const v2u *>(rec + 16);
  return v8i{(int)x.x, (int)x.y, (int)y.x, (int)y.y, (int)z.x, (int)z.y, 0, 0};
}

template <int NW, int MBLK, int CNT>
__global__ __launch_bounds__(NW * 64) void f6d(P p) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  float (*part)[MBLK][64][4] = reinterpret_cast<float (*)[MBLK][64][4]>(lds_raw);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane & 15, kb = lane >> 4;
  const int xq = (int)gridDim.x >> 3, xr = (int)gridDim.x & 7, xx = blockIdx.x & 7;
  const int n0 = (xx * xq + min(xx, xr) + ((int)blockIdx.x >> 3)) * 16;
  const int G = p.G;
  const int per = (G + 1 + NW - 1) / NW;
  const int i0 = wave * per, i1 = min(i0 + per, G + 1);
  const bool keeper = i1 == G + 1 && i0 <= G;
  const int ng = min(i1, G) - i0;

  // the whole weight slice of this wave, in flight from the first instructions
  v8i w[CNT];
  v4f sb[CNT];
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    w[j] = v8i{0, 0, 0, 0, 0, 0, 0, 0};
    sb[j] = v4f{0.f, 0.f, 0.f, 0.f};
    if (j < ng) {
      w[j] = load_frag(p.B6 + ((int64_t)(i0 + j) * p.Npad + n0 + row) * PITCH + kb * 24);
      sb[j] = *reinterpret_cast<const v4f *>(p.sB32 + (int64_t)(i0 + j) * p.Npad + n0 + 4 * kb);
    }
  }
  v4i wk[2] = {};
  v2u sbk = {0u, 0u};
  if (keeper) {
    const int8_t *kp = p.B8 + (int64_t)(n0 + row) * 128 + kb * 16;
    wk[0] = *reinterpret_cast<const v4i *>(kp);
    wk[1] = *reinterpret_cast<const v4i *>(kp + 64);
    sbk = *reinterpret_cast<const v2u *>(p.sB8 + n0 + 4 * kb);
  }
  v8i a[CNT];
  float sa[CNT];
  auto load_act = [&](int tb, int j) {
    const int m = min(tb * 16 + row, p.M - 1);
    const uint8_t *rec = p.A6 + ((int64_t)(i0 + j) * p.Mpad + m) * PITCH;
    a[j] = load_frag(rec + kb * 24);
    sa[j] = *reinterpret_cast<const float *>(rec + 100);
  };
#pragma unroll
  for (int j = 0; j < CNT; ++j)
    if (j < ng) load_act(0, j);
  v4i ak[2] = {};
  unsigned sak = 0;
  auto load_keeper_act = [&](int tb) {
    const int m = min(tb * 16 + row, p.M - 1);
    const int8_t *kp = p.A8 + (int64_t)m * 128 + kb * 16;
    ak[0] = *reinterpret_cast<const v4i *>(kp);
    ak[1] = *reinterpret_cast<const v4i *>(kp + 64);
    sak = *reinterpret_cast<const unsigned short *>(p.sA8 + m);
  };
  if (keeper) load_keeper_act(0);

  float c[MBLK][4];
#pragma unroll
  for (int tb = 0; tb < MBLK; ++tb)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[tb][r] = 0.f;

#pragma unroll
  for (int tb = 0; tb < MBLK; ++tb) {
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
      __builtin_amdgcn_sched_barrier(0);
      if (j < ng) {
        const v4f acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w[j], a[j], v4f{0.f, 0.f, 0.f, 0.f}, 3, 3, 0, 0, 0, 0);
        const float s = sa[j];
        if (tb + 1 < MBLK) load_act(tb + 1, j);
        float t[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = acc[r] * s;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[tb][r] = __builtin_fmaf(t[r], sb[j][r], c[tb][r]);
      }
    }
    if (keeper) {
      v4i acc = {0, 0, 0, 0};
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wk[0], ak[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wk[1], ak[1], acc, 0, 0, 0);
      const float s8 = (float)__builtin_bit_cast(half_t, (unsigned short)sak);
      const half_t *hv = reinterpret_cast<const half_t *>(&sbk);
      if (tb + 1 < MBLK) load_keeper_act(tb + 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) c[tb][r] = __builtin_fmaf((float)acc[r] * s8, (float)hv[r], c[tb][r]);
    }
  }
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
    if (m < p.M) {
      v2u o;
      half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = (half_t)s[r];
      *reinterpret_cast<v2u *>(p.D + (int64_t)m * p.N + n0 + 4 * kb) = o;
    }
  }
}

typedef void (*Launch)(P, hipStream_t);
template <int NW, int MBLK, int CNT> static void l_f6d(P p, hipStream_t s) {
  const size_t lds = (size_t)NW * MBLK * 64 * 16;
  static bool done = false;
  if (!done) { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&f6d<NW, MBLK, CNT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
  hipLaunchKernelGGL((f6d<NW, MBLK, CNT>), dim3(p.N / 16), dim3(NW * 64), lds, s, p);
}

static float graph_us(Launch f, std::vector<P> &ps, int nsets, int iters, hipStream_t st) {
  hipGraph_t g; hipGraphExec_t ge;
  f(ps[0], st); CK(hipStreamSynchronize(st));
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < iters; ++i) f(ps[i % nsets], st);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, st); CK(hipGraphLaunch(ge, st)); hipEventRecord(e1, st); CK(hipStreamSynchronize(st));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms / iters * 1e3f < best) best = ms / iters * 1e3f;
  }
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return best;
}

static void fill(void *d, size_t n, uint32_t seed) {
  std::vector<uint32_t> h((n + 3) / 4);
  uint32_t s = seed;
  for (auto &x : h) { s = s * 1664525u + 1013904223u; x = s & 0x3f3f3f3fu; }   // small positive BF6-ish fields / finite floats
  CK(hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice));
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int shapes[][2] = {{4096, 4096}, {13824, 5120}, {5120, 5120}};
  for (auto &sh : shapes) {
    const int N = sh[0], K = sh[1], G = (K - 128) / 128, Npad = (N + 255) / 256 * 256;
    const size_t wbytes = (size_t)G * Npad * PITCH + (size_t)G * Npad * 4 + (size_t)N * 128 + N * 2;
    const int nsets = (int)((600u << 20) / wbytes) + 1;
    for (int M : {16, 64, 128, 256}) {
      const int Mpad = (M + 255) / 256 * 256;
      uint8_t *A6; int8_t *A8; half_t *sA8, *D;
      CK(hipMalloc(&A6, (size_t)G * Mpad * PITCH)); CK(hipMalloc(&A8, (size_t)M * 128)); CK(hipMalloc(&sA8, M * 2)); CK(hipMalloc(&D, (size_t)M * N * 2));
      fill(A6, (size_t)G * Mpad * PITCH, 1); fill(A8, (size_t)M * 128, 2); fill(sA8, M * 2, 3);
      std::vector<P> ps(nsets);
      std::vector<void *> frees;
      for (int i = 0; i < nsets; ++i) {
        uint8_t *B6; float *sB32; int8_t *B8; half_t *sB8;
        CK(hipMalloc(&B6, (size_t)G * Npad * PITCH)); CK(hipMalloc(&sB32, (size_t)G * Npad * 4)); CK(hipMalloc(&B8, (size_t)N * 128)); CK(hipMalloc(&sB8, N * 2));
        if (i == 0) { fill(B6, (size_t)G * Npad * PITCH, 4); fill(sB32, (size_t)G * Npad * 4, 5); fill(B8, (size_t)N * 128, 6); fill(sB8, N * 2, 7); }
        else { CK(hipMemcpy(B6, ps[0].B6, (size_t)G * Npad * PITCH, hipMemcpyDeviceToDevice)); CK(hipMemcpy(sB32, ps[0].sB32, (size_t)G * Npad * 4, hipMemcpyDeviceToDevice));
               CK(hipMemcpy(B8, ps[0].B8, (size_t)N * 128, hipMemcpyDeviceToDevice)); CK(hipMemcpy(sB8, ps[0].sB8, N * 2, hipMemcpyDeviceToDevice)); }
        ps[i] = P{A6, B6, sB32, A8, B8, sA8, sB8, D, M, N, G, Mpad, Npad};
        frees.push_back(B6); frees.push_back(sB32); frees.push_back(B8); frees.push_back(sB8);
      }
      const int per = (G + 1 + 7) / 8;
      Launch f = nullptr;
      const int mblk = (M + 15) / 16;
#define PICK(MB) (per <= 4 ? (Launch)l_f6d<8, MB, 4> : (per <= 5 ? (Launch)l_f6d<8, MB, 5> : (Launch)l_f6d<8, MB, 8>))
      if (mblk <= 1) f = PICK(1); else if (mblk <= 4) f = PICK(4); else if (mblk <= 8) f = PICK(8); else f = PICK(16);
      int iters = nsets * 2 > 64 ? nsets * 2 : 64; iters -= iters % nsets;
      const float hot = graph_us(f, ps, 1, iters, st), cold = graph_us(f, ps, nsets, iters, st);
      const double ops = 2.0 * M * N * K;
      printf("f6d %4d x %5d x %5d  (items/wave %d)  hot %6.2f us  cold %6.2f us  %6.1f TOPS cold   weights %.1f MB\n", M, N, K, per, hot, cold, ops / cold / 1e6, wbytes / 1e6);
      for (void *q : frees) hipFree(q);
      hipFree(A6); hipFree(A8); hipFree(sA8); hipFree(D);
    }
  }
  return 0;
}
