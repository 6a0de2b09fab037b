// Probe: issue cost of the VALU ops the W4A4 dequant/widen path uses, alone and beside MFMAs, 1 vs 2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
  float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0001f, c = 0.5f;
  float e0 = 0, e1 = 1, e2 = 2, e3 = 3, e4 = 4, e5 = 5, e6 = 6, e7 = 7;
  unsigned u0 = threadIdx.x, u1 = 3, u2 = 5, u3 = 7;
  v4i fa = {1, 2, 3, 4}, fb = {5, 6, 7, 8};
  v16i acc = {0}, acc2 = {0};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 16 v_fma_f32
      asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (MODE == 1) {  // 16 v_pk_fma_f32
      asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n") : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&e0), "v"(*(double*)&e2));
    } else if (MODE == 2) {  // 16 v_fma_mix_f32 (f16 src1)
      asm volatile(REP8("v_fma_mix_f32 %0, %0, %8, %9 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %1, %8, %9 op_sel_hi:[0,1,0]\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (MODE == 3) {  // 8 shift + 8 and
      asm volatile(REP8("v_lshlrev_b32 %0, 4, %1\n v_and_b32 %2, 0xf0f0f0f0, %3\n") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
    } else if (MODE == 4) {  // 4 MFMA only (2 independent chains)
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc2, 0, 0, 0);
    } else if (MODE == 5) {  // 4 MFMA, ONE dependent chain
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
    } else if (MODE == 6 || MODE == 7 || MODE == 8) {  // 4 MFMA (2 chains) interleaved with 8*VPM fma each
      constexpr int VPM = MODE == 6 ? 1 : (MODE == 7 ? 2 : 3);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j & 1) acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc2, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < VPM; ++q)
          asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      }
    } else if (MODE == 9) {  // 4 MFMA (2 chains) + 8 pk_fma each
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j & 1) acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc2, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
        asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %5\n") : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&e0), "v"(*(double*)&e2));
      }
    }
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7 + u0 + u1 + u2 + u3;
  for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int threads, float *out, int per_iter_valu, int per_iter_mfma) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ns_per_iter = ms * 1e6 / iters;
  int wps = threads / 256;
  printf("%-34s waves/SIMD %d: %.1f ns/iter/wave-set -> per SIMD: %.2f ns per VALU, %.2f ns per MFMA\n", name, wps, ns_per_iter,
         per_iter_valu ? ns_per_iter / (per_iter_valu * wps) : 0.0, per_iter_mfma ? ns_per_iter / (per_iter_mfma * wps) : 0.0);
}
int main() {
  float *out; hipMalloc(&out, 256 * 512 * 4);
  for (int t = 256; t <= 512; t += 256) {
    run<0>("16 v_fma_f32", t, out, 16, 0);
    run<1>("16 v_pk_fma_f32", t, out, 16, 0);
    run<2>("16 v_fma_mix_f32", t, out, 16, 0);
    run<3>("8 lshl + 8 and", t, out, 16, 0);
    run<4>("4 MFMA 2 chains", t, out, 0, 4);
    run<5>("4 MFMA 1 chain", t, out, 0, 4);
    run<6>("4 MFMA + 32 fma", t, out, 32, 4);
    run<7>("4 MFMA + 64 fma", t, out, 64, 4);
    run<8>("4 MFMA + 96 fma", t, out, 96, 4);
    run<9>("4 MFMA + 32 pk_fma", t, out, 32, 4);
  }
  return 0;
}
