// Probe for the block-scaled MFMA on gfx950 with BF6 (E3M2) operands holding small integers:
//  (1) operand packing (32 six-bit values per lane in 6 VGPRs) and exactness of integer dot products with unit scales
//  (2) issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 (bf6 x bf6) vs v_mfma_i32_32x32x32_i8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

static unsigned bf6(int v) {   // E3M2, bias 3: integers -8..8
  static const unsigned mag[9] = {0x00, 0x0C, 0x10, 0x12, 0x14, 0x15, 0x16, 0x17, 0x18};
  return (v < 0 ? 0x20u : 0u) | mag[v < 0 ? -v : v];
}

__global__ void one_mfma(const unsigned *A, const unsigned *B, float *D, int fmt) {
  const int lane = threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = A[lane * 8 + i]; b[i] = B[lane * 8 + i]; }
  v16f c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  // cbsz / blgp = 3: BF6 (E3M2) for A / B; scales: E8M0 127 = 2^0
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 3, 3, 0, 127, 0, 127);
  for (int r = 0; r < 16; ++r) D[lane * 16 + r] = c[r];
}

template <int MODE>
__global__ void rate_kernel(float *out, int iters) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 7 + i; b[i] = threadIdx.x * 3 + i; }
  if (MODE == 0) {
    v16f c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = c2[i] = c3[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 3, 3, 0, 127, 0, 127);
      c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 3, 3, 0, 127, 0, 127);
      c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 3, 3, 0, 127, 0, 127);
      c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 3, 3, 0, 127, 0, 127);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    v4i a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    v16i c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = c2[i] = c3[i] = 0;
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(c0[0] + c1[1] + c2[2] + c3[3]);
  }
}

typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void rate16(float *out, int iters) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 7 + i; b[i] = threadIdx.x * 3 + i; }
  v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 3, 3, 0, 127, 0, 127);
    c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 3, 3, 0, 127, 0, 127);
    c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 3, 3, 0, 127, 0, 127);
    c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 3, 3, 0, 127, 0, 127);
    c4 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c4, 3, 3, 0, 127, 0, 127);
    c5 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c5, 3, 3, 0, 127, 0, 127);
    c6 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c6, 3, 3, 0, 127, 0, 127);
    c7 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c7, 3, 3, 0, 127, 0, 127);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
}

int main() {
  std::mt19937 rng(1);
  std::vector<int> Am(32 * 64), Bm(32 * 64);            // A[i][k], B[j][k], values -8..7 (and 8 for coverage)
  for (auto &v : Am) v = (int)(rng() % 16) - 8;
  for (auto &v : Bm) v = (int)(rng() % 17) - 8;
  // lane l holds row l%32, k in [32*(l/32), +32): 32 six-bit fields, little-endian bit stream over 6 dwords
  auto pack = [&](const std::vector<int> &M, std::vector<unsigned> &out) {
    out.assign(64 * 8, 0);
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 32; ++e) {
        const unsigned code = bf6(M[(l % 32) * 64 + 32 * (l / 32) + e]);
        const int bit = 6 * e;
        out[l * 8 + bit / 32] |= code << (bit % 32);
        if (bit % 32 > 26) out[l * 8 + bit / 32 + 1] |= code >> (32 - bit % 32);
      }
  };
  std::vector<unsigned> Ap, Bp;
  pack(Am, Ap); pack(Bm, Bp);
  unsigned *dA, *dB; float *dD;
  hipMalloc(&dA, Ap.size() * 4); hipMalloc(&dB, Bp.size() * 4); hipMalloc(&dD, 64 * 16 * 4);
  hipMemcpy(dA, Ap.data(), Ap.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, Bp.data(), Bp.size() * 4, hipMemcpyHostToDevice);
  one_mfma<<<1, 64>>>(dA, dB, dD, 3);
  std::vector<float> D(64 * 16);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0; double maxerr = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r) {
      const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      // which operand is rows / cols: try D[row][col] = sum_k A[row][k] * B[col][k]
      long ref = 0;
      for (int k = 0; k < 64; ++k) ref += (long)Am[row * 64 + k] * Bm[col * 64 + k];
      const double e = fabs((double)D[l * 16 + r] - (double)ref);
      if (e > maxerr) maxerr = e;
      if (e != 0) ++bad;
    }
  printf("PROBE bf6 one MFMA (D[row][col] = A[row].B[col]): %d mismatches of 1024, max abs err %.3f\n", bad, maxerr);
  if (bad) {
    int bad2 = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) {
        const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        long ref = 0;
        for (int k = 0; k < 64; ++k) ref += (long)Am[col * 64 + k] * Bm[row * 64 + k];
        if ((double)D[l * 16 + r] != (double)ref) ++bad2;
      }
    printf("PROBE bf6 transposed convention: %d mismatches\n", bad2);
    printf("  sample D[0..3] = %.1f %.1f %.1f %.1f\n", D[0], D[1], D[2], D[3]);
  }
  // rate: one wave per SIMD (256 threads per WG, 1 WG per CU), 4 independent chains
  float *dout; hipMalloc(&dout, 256 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) rate_kernel<0><<<256, 256>>>(dout, iters); else rate_kernel<1><<<256, 256>>>(dout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = 4.0 * iters;                       // per wave
    const double ns_per = ms * 1e6 / mfmas;
    const double ops = 256.0 * 4 * mfmas * 2.0 * 32 * 32 * (mode == 0 ? 64 : 32);
    printf("PROBE rate %s: %.2f ns per MFMA per SIMD, %.0f Tops/s chip\n", mode == 0 ? "bf6 32x32x64 scaled" : "i8 32x32x32", ns_per, ops / (ms * 1e-3) / 1e12);
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    rate16<<<256, 256>>>(dout, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  {
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = 8.0 * iters;
    printf("PROBE rate bf6 16x16x128 scaled: %.2f ns per MFMA per SIMD, %.0f Tops/s chip\n", ms * 1e6 / mfmas,
           256.0 * 4 * mfmas * 2.0 * 16 * 16 * 128 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
