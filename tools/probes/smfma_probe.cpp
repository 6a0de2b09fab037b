// Probe (round 5): can the matrix pipe form the de-quantisation's scale products?  s[m][n] = sA[m] * sB[n] is a rank-1 matrix, i.e. ONE
// f16 MFMA with a single non-zero k: v_mfma_f32_32x32x8_f16 / v_mfma_f32_16x16x16_f16 (legacy shapes) and the gfx950 double-K forms.
// Measures (a) exactness of the products -- fp16 x fp16 is exact in FP32; subnormal fp16 inputs included -- and (b) the issue cost of each
// shape, back to back, one and two waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O2 tools/probes/smfma_probe.cpp -o build/tools/smfma_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 half_t;
typedef half_t v4h __attribute__((ext_vector_type(4)));
typedef half_t v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// outer product through 32x32x8: A[i][k] from lane i + 32 (k / 4), element k % 4; B[k][j] from lane j + 32 (k / 4); only k = 0 non-zero
__global__ void outer32(const half_t *sb, const half_t *sa, float *out) {
  const int l = threadIdx.x;
  v4h a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  if (l < 32) { a[0] = sb[l]; b[0] = sa[l]; }
  v16f c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void outer16(const half_t *sb, const half_t *sa, float *out) {
  const int l = threadIdx.x;
  v4h a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  if (l < 16) { a[0] = sb[l]; b[0] = sa[l]; }
  v4f c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

template <int KIND>
__global__ __launch_bounds__(512) void rate(float *out, int iters, unsigned long long *cyc) {
  v4h a4 = {(half_t)1.f, (half_t)0.f, (half_t)0.f, (half_t)0.f}, b4 = {(half_t)(1.f + threadIdx.x * 0.001f), (half_t)0.f, (half_t)0.f, (half_t)0.f};
  v8h a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = i ? (half_t)0.f : (half_t)1.f; b8[i] = i ? (half_t)0.f : b4[0]; }
  v16f c0, c1, c2, c3;
  for (int i = 0; i < 16; ++i) c0[i] = c1[i] = c2[i] = c3[i] = 0.f;
  v4f d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c3, 0, 0, 0);
    } else if constexpr (KIND == 1) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c3, 0, 0, 0);
    } else if constexpr (KIND == 2) {
      d0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d3, 0, 0, 0);
    } else {
      d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, d3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 7 && threadIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[1] + d2[2] + d3[3];
}

int main() {
  // ---- exactness
  std::vector<half_t> sb(32), sa(32);
  const uint16_t specials[] = {0x0001, 0x0002, 0x03FF, 0x0400, 0x0401, 0x7BFF, 0x3C00, 0x3555, 0x2E66, 0x1234, 0x00FF, 0x8001, 0xBC01, 0x0018, 0x7800, 0x4000};
  for (int i = 0; i < 32; ++i) {
    uint16_t u = i < 16 ? specials[i] : (uint16_t)(0x1000 + 977 * i);
    uint16_t v = i < 16 ? specials[15 - i] : (uint16_t)(0x2000 + 1543 * i);
    memcpy(&sb[i], &u, 2); memcpy(&sa[i], &v, 2);
  }
  half_t *dsb, *dsa; float *dout;
  CK(hipMalloc(&dsb, 64)); CK(hipMalloc(&dsa, 64)); CK(hipMalloc(&dout, 32 * 32 * 4));
  CK(hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice));
  std::vector<float> out(1024);
  hipLaunchKernelGGL(outer32, dim3(1), dim3(64), 0, 0, dsb, dsa, dout);
  CK(hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    const float e = (float)sb[i] * (float)sa[j];
    if (memcmp(&e, &out[i * 32 + j], 4) != 0) { if (bad < 5) printf("  32x32x8 mismatch [%d][%d]: got %a want %a\n", i, j, out[i * 32 + j], e); ++bad; }
  }
  printf("outer product on v_mfma_f32_32x32x8_f16 (fp16 incl. subnormals / extremes): %d of 1024 products differ from the exact FP32 product\n", bad);
  hipLaunchKernelGGL(outer16, dim3(1), dim3(64), 0, 0, dsb, dsa, dout);
  CK(hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost));
  bad = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    const float e = (float)sb[i] * (float)sa[j];
    if (memcmp(&e, &out[i * 16 + j], 4) != 0) ++bad;
  }
  printf("outer product on v_mfma_f32_16x16x16_f16: %d of 256 differ\n", bad);
  // ---- rate
  float *big; unsigned long long *cyc;
  CK(hipMalloc(&big, 256 * 512 * 4)); CK(hipMalloc(&cyc, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *names[4] = {"v_mfma_f32_32x32x8_f16 (legacy)", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x16_f16 (legacy)", "v_mfma_f32_16x16x32_f16"};
  void (*fns[4])(float *, int, unsigned long long *) = {rate<0>, rate<1>, rate<2>, rate<3>};
  const int iters = 20000;
  for (int k = 0; k < 4; ++k)
    for (int w = 1; w <= 2; ++w) {
      hipLaunchKernelGGL(fns[k], dim3(256), dim3(256 * w), 0, 0, big, 2000, cyc);
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(fns[k], dim3(256), dim3(256 * w), 0, 0, big, iters, cyc);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long hc; CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
      printf("%-34s %d wave(s)/SIMD: %6.2f ns per MFMA per SIMD, %5.1f s_memtime cycles per MFMA per wave\n", names[k], w, ms * 1e6 / iters / 4 / w,
             (double)hc / iters / 4);
    }
  return 0;
}
