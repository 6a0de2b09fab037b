// Native tuning harness for libatom_hip.so (no Python): correctness against a naive GPU kernel + hipEvent timing.
//   make -C atom_amd/csrc tools   (builds build/tools/gemm_bench against the -DATOM_TOOLS library, which reads ATOM_F6_CFG etc.)
//   build/tools/gemm_bench M N K iters [check_rows]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>
#include "../include/atom_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef _Float16 half_t;

__device__ inline int nib(const uint8_t *p, int k) {
  int v = (p[k >> 1] >> ((k & 1) * 4)) & 0xF;
  return v >= 8 ? v - 16 : v;
}

// one thread per output element, straight from the definition, FP64 accumulate
__global__ void naive_ref(const uint8_t *A4, const uint8_t *B4, const half_t *sA, const half_t *sB, const int8_t *A8,
                          const int8_t *B8, const half_t *sA8, const half_t *sB8, double *D, int M, int N, int K4, int ldA) {
  int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N || m >= M) return;
  int G = K4 / 128;
  double acc = 0;
  for (int g = 0; g < G; ++g) {
    int s = 0;
    for (int k = 0; k < 128; ++k) s += nib(A4 + (size_t)m * (K4 / 2), g * 128 + k) * nib(B4 + (size_t)n * (K4 / 2), g * 128 + k);
    acc += (double)s * (double)(float)sA[(size_t)g * ldA + m] * (double)(float)sB[(size_t)g * N + n];
  }
  int s = 0;
  for (int k = 0; k < 128; ++k) s += (int)A8[(size_t)m * 128 + k] * (int)B8[(size_t)n * 128 + k];
  acc += (double)s * (double)(float)sA8[m] * (double)(float)sB8[n];
  D[(size_t)m * N + n] = acc;
}

int main(int argc, char **argv) {
  int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  int iters = argc > 4 ? atoi(argv[4]) : 50;
  int check_rows = argc > 5 ? atoi(argv[5]) : 256;
  int K4 = K - 128, G = K4 / 128;
  std::mt19937_64 rng(123);
  auto fill_u8 = [&](std::vector<uint8_t> &v) { for (auto &x : v) x = (uint8_t)(rng() & 0xFF); };
  auto fill_sc = [&](std::vector<half_t> &v) { for (auto &x : v) x = (half_t)(0.005f + 0.045f * (float)((rng() >> 11) * (1.0 / 9007199254740992.0))); };
  std::vector<uint8_t> hA4((size_t)M * K4 / 2), hB4((size_t)N * K4 / 2), hA8((size_t)M * 128), hB8((size_t)N * 128);
  std::vector<half_t> hsA((size_t)G * M), hsB((size_t)G * N), hsA8(M), hsB8(N);
  fill_u8(hA4); fill_u8(hB4); fill_u8(hA8); fill_u8(hB8);
  if (const char *dm = getenv("ATOM_DATA")) {   // power / toggle sensitivity: zero = all codes 0, const = every code 1, small = codes in [-1, 1]
    const int mode = !strcmp(dm, "zero") ? 0 : (!strcmp(dm, "const") ? 1 : 2);
    for (auto *v : {&hA4, &hB4, &hA8, &hB8})
      for (auto &x : *v) x = mode == 0 ? 0 : (mode == 1 ? 0x11 : (uint8_t)(((x & 1) ? 0x01 : 0x0F) | ((x & 2) ? 0x10 : 0xF0)));
  }
  fill_sc(hsA); fill_sc(hsB); fill_sc(hsA8); fill_sc(hsB8);
  for (int g = 0; g < G; ++g) for (int n = 0; n < N; n += 2) hsB[(size_t)g * N + n + 1] = hsB[(size_t)g * N + n];
  void *A4, *B4, *A8, *B8, *sA, *sB, *sA8, *sB8, *D;
  auto up = [&](void **d, const void *h, size_t bytes) { CK(hipMalloc(d, bytes)); CK(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice)); };
  up(&A4, hA4.data(), hA4.size()); up(&B4, hB4.data(), hB4.size()); up(&A8, hA8.data(), hA8.size()); up(&B8, hB8.data(), hB8.size());
  up(&sA, hsA.data(), hsA.size() * 2); up(&sB, hsB.data(), hsB.size() * 2); up(&sA8, hsA8.data(), hsA8.size() * 2); up(&sB8, hsB8.data(), hsB8.size() * 2);
  CK(hipMalloc(&D, (size_t)M * N * 2));
  CK(hipMemset(D, 0xFF, (size_t)M * N * 2));

  // ATOM_AWIDE=1: feed the GEMM the wide activation format (int8 code*16, even/odd de-interleaved per 32 channels)
  const bool awide = getenv("ATOM_AWIDE") != nullptr;
  void *Ain = A4;
  const int layout = ATOM_SCALE_LAYOUT_PLAIN | (awide ? ATOM_A_WIDE : 0);
  if (awide) {
    std::vector<uint8_t> hw((size_t)M * K4);
    for (size_t m = 0; m < (size_t)M; ++m)
      for (int c = 0; c < K4 / 32; ++c)
        for (int j = 0; j < 16; ++j) {
          const uint8_t b = hA4[m * (K4 / 2) + c * 16 + j];        // channels 32c+2j (low nibble), 32c+2j+1 (high)
          hw[m * K4 + c * 32 + j] = (uint8_t)(b << 4);
          hw[m * K4 + c * 32 + 16 + j] = (uint8_t)(b & 0xF0);
        }
    up(&Ain, hw.data(), hw.size());
  }
  // ATOM_F6=1: feed the GEMM the BF6 group-major format (both operands)
  void *Bin = B4;
  int layout2 = layout;
  if (getenv("ATOM_F6")) {
    static const uint8_t mag[9] = {0x00, 0x0C, 0x10, 0x12, 0x14, 0x15, 0x16, 0x17, 0x18};
    auto code6 = [&](int v) { return (uint8_t)((v < 0 ? 0x20 : 0) | mag[v < 0 ? -v : v]); };
    auto conv = [&](const std::vector<uint8_t> &packed, int rows, const half_t *scales /* [G][rows] or null */) {
      const int rp = (rows + 255) / 256 * 256;
      std::vector<uint8_t> out((size_t)G * rp * 104, 0);
      for (int g = 0; g < G; ++g)
        for (int r = 0; r < rows; ++r) {
          uint8_t *dst = &out[((size_t)g * rp + r) * 104];
          for (int e = 0; e < 128; ++e) {
            const int kk = g * 128 + e;
            const uint8_t pb = packed[(size_t)r * (K4 / 2) + kk / 2];
            const int nv = (kk & 1) ? (int8_t)(pb & 0xF0) >> 4 : (int8_t)(pb << 4) >> 4;
            const unsigned c = code6(nv);
            const int bit = 6 * e;
            dst[bit / 8] |= (uint8_t)(c << (bit % 8));
            if (bit % 8 > 2) dst[bit / 8 + 1] |= (uint8_t)(c >> (8 - bit % 8));
          }
          if (scales) { memcpy(dst + 96, &scales[(size_t)g * rows + r], 2); const float f = (float)scales[(size_t)g * rows + r]; memcpy(dst + 100, &f, 4); }
        }
      return out;
    };
    std::vector<uint8_t> ha = conv(hA4, M, hsA.data());
    up(&Ain, ha.data(), ha.size());
    // weights through the library's own repack: codes + float32 scales (ATOM_B_F6S) unless ATOM_NO_F6S is set
    const bool f6s = !getenv("ATOM_NO_F6S");
    CK(hipMalloc(&Bin, atom_f6_weight_bytes(N, K)));
    { int st = f6s ? atom_repack_weight_f6s(B4, sB, N, K, Bin, nullptr) : atom_repack_weight_f6(B4, N, K, Bin, nullptr);
      if (st) { printf("repack: %s\n", atom_strerror(st)); return 1; } }
    layout2 = ATOM_SCALE_LAYOUT_PLAIN | ATOM_AB_F6 | (f6s ? ATOM_B_F6S : 0);
  }
#ifdef ATOM_B_SCALE_PAIRS
  if (!getenv("ATOM_NO_PAIRS")) layout2 |= ATOM_B_SCALE_PAIRS;   // hsB above: channel pairs share their scales (ATOM_NO_PAIRS: the generic kernels)
#endif
  size_t wsb = getenv("ATOM_WS") ? atom_gemm_w4a4_workspace_bytes(M, N, K) : 0;
  void *ws = nullptr; if (wsb) CK(hipMalloc(&ws, wsb));
  printf("workspace bytes %zu\n", wsb);
#define atom_gemm_w4a4_f16(a, b, c, d, e, f, g, h, i, j, k, l, m, n, o, p) atom_gemm_w4a4_f16_ws(a, b, c, d, e, f, g, h, i, j, k, l, m, n, o, ws, wsb, p)
  int st = atom_gemm_w4a4_f16(Ain, Bin, sA, sB, A8, B8, sA8, sB8, D, M, N, K, 128, 128, layout2, nullptr);
  if (st) { printf("atom_gemm_w4a4_f16: %s\n", atom_strerror(st)); return 1; }
  CK(hipDeviceSynchronize());

  // correctness on the first and last check_rows rows
  int cr = check_rows < M ? check_rows : M;
  if (cr > 0) {
    double *Dref; CK(hipMalloc(&Dref, (size_t)M * N * 8));
    hipLaunchKernelGGL(naive_ref, dim3((N + 255) / 256, M), dim3(256), 0, 0, (const uint8_t *)A4, (const uint8_t *)B4,
                       (const half_t *)sA, (const half_t *)sB, (const int8_t *)A8, (const int8_t *)B8, (const half_t *)sA8,
                       (const half_t *)sB8, Dref, (cr == M ? M : M), N, K4, M);
    CK(hipDeviceSynchronize());
    std::vector<double> href((size_t)M * N); std::vector<half_t> hD((size_t)M * N);
    CK(hipMemcpy(href.data(), Dref, href.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hD.data(), D, hD.size() * 2, hipMemcpyDeviceToHost));
    double ss = 0; for (double v : href) ss += v * v; double rms = sqrt(ss / href.size());
    double maxerr = 0; size_t bad = 0;
    for (size_t i = 0; i < href.size(); ++i) {
      double e = fabs((double)(float)hD[i] - href[i]);
      if (e > maxerr) maxerr = e;
      if (e > 1e-3 * fabs(href[i]) + 1e-3 * rms) ++bad;
    }
    printf("check %dx%dx%d: rms %.4g max_abs_err %.4g bad %zu / %zu -> %s\n", M, N, K, rms, maxerr, bad, href.size(), bad ? "FAIL" : "ok");
    CK(hipFree(Dref));
  }

  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (getenv("ATOM_WS_CACHED")) layout2 |= ATOM_WS_WEIGHT_CACHED;   // the weight's re-coded form is in `ws` since the call above
  for (int i = 0; i < 5; ++i) atom_gemm_w4a4_f16(Ain, Bin, sA, sB, A8, B8, sA8, sB8, D, M, N, K, 128, 128, layout2, nullptr);
  CK(hipDeviceSynchronize());
  float best = 1e30f, tot = 0;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) atom_gemm_w4a4_f16(Ain, Bin, sA, sB, A8, B8, sA8, sB8, D, M, N, K, 128, 128, layout2, nullptr);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters; tot += ms; if (ms < best) best = ms;
  }
  double ops = 2.0 * M * N * K;
  const char *var = getenv("ATOM_GEMM_VARIANT");
  printf("RESULT variant=%s M=%d N=%d K=%d  avg %.2f us  best %.2f us  %.1f TOPS (avg)  %.1f TOPS (best)  frac_of_5033=%.3f\n",
         var ? var : "default", M, N, K, tot / 5 * 1e3, best * 1e3, ops / (tot / 5 * 1e-3) / 1e12, ops / (best * 1e-3) / 1e12,
         ops / (tot / 5 * 1e-3) / 1e12 / 5033.0);
  return 0;
}
