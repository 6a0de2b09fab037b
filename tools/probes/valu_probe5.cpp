// Probe 5: does VALU work hide under the block-scaled BF6 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, cbsz = blgp = 3)?
// Hand-interleaved streams (inline asm), 1 and 2 waves per SIMD, all 256 CUs busy.  s_memtime ticks = shader cycles.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/valu_probe5.cpp -o build/valu_probe5
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define F(n) "v_fma_f32 %" #n ", %" #n ", %18, %19\n"
#define V4(OP) OP(2) OP(3) OP(4) OP(5)
#define V8(OP) V4(OP) OP(6) OP(7) OP(8) OP(9)
#define V12(OP) V8(OP) OP(10) OP(11) OP(12) OP(13)
#define V16(OP) V12(OP) OP(14) OP(15) OP(16) OP(17)
// accumulate form (C = D) and the C = 0 form the GEMM uses for the first MFMA of a group
#define MA "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %20, %21, %0, %22, %22 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n"
#define MB "v_mfma_scale_f32_32x32x64_f8f6f4 %1, %20, %21, %1, %22, %22 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n"
#define ZA "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %20, %21, 0, %22, %22 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n"
#define ZB "v_mfma_scale_f32_32x32x64_f8f6f4 %1, %20, %21, 0, %22, %22 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n"
#define OPS : "+v"(acc), "+v"(acc2), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
              "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
            : "v"(b), "v"(c), "v"(fa), "v"(fb), "v"(sc)
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, unsigned long long *cyc) {
  float a[16]; for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
  float b = 1.0001f, c = 0.5f;
  v6i fa = {0x0c30c30c, 0x30c30c30, (int)0xc30c30c3, 0x0c30c30c, 0x30c30c30, (int)threadIdx.x}, fb = fa;
  int sc = 127;
  v16f acc = {0}, acc2 = {1};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) asm volatile(MA MB MA MB OPS);
    if (MODE == 1) asm volatile(MA V4(F) MB V4(F) MA V4(F) MB V4(F) OPS);
    if (MODE == 2) asm volatile(MA V8(F) MB V8(F) MA V8(F) MB V8(F) OPS);
    if (MODE == 3) asm volatile(MA V12(F) MB V12(F) MA V12(F) MB V12(F) OPS);
    if (MODE == 4) asm volatile(MA V16(F) MB V16(F) MA V16(F) MB V16(F) OPS);
    if (MODE == 5) asm volatile(MA V16(F) V8(F) MB V16(F) V8(F) MA V16(F) V8(F) MB V16(F) V8(F) OPS);
    if (MODE == 6) asm volatile(V16(F) V16(F) V16(F) V16(F) OPS);                 // 64 fma, no MFMA
    if (MODE == 7) asm volatile(ZA MA ZB MB OPS);                                  // the GEMM's chain shape: (C=0, accumulate) x 2
    if (MODE == 8) asm volatile(ZA MA V16(F) V16(F) ZB MB V16(F) V16(F) OPS);      // chain pair then 32 fma (phase-separated)
    if (MODE == 9) asm volatile(ZA V16(F) MA V16(F) ZB V16(F) MB V16(F) OPS);      // same work, interleaved
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 7 && threadIdx.x == 0) cyc[0] = t1 - t0;
  float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + acc[i] + acc2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int threads, float *out, int valu, int mfma) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  static unsigned long long *cyc = nullptr; if (!cyc) hipMalloc(&cyc, 8);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 2000, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ns = ms * 1e6 / iters; int wps = threads / 256;
  unsigned long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  double cpi = (double)hc / iters;
  printf("%-34s w/SIMD %d: %7.1f ns/iter %7.1f cyc/iter (%.2f GHz) | per SIMD per MFMA: %6.2f ns %6.1f cyc", name, wps, ns, cpi, cpi / ns,
         mfma ? ns / (mfma * wps) : 0.0, mfma ? cpi / (mfma * wps) : 0.0);
  if (valu) printf(" | per VALU %5.2f cyc", cpi / (valu * wps));
  printf("\n");
}
int main() {
  float *out; hipMalloc(&out, 256 * 512 * 4);
  for (int t = 256; t <= 512; t += 256) {
    run<0>("4 MFMA bf6", t, out, 0, 4);
    run<1>("4x(MFMA + 4 fma)", t, out, 16, 4);
    run<2>("4x(MFMA + 8 fma)", t, out, 32, 4);
    run<3>("4x(MFMA + 12 fma)", t, out, 48, 4);
    run<4>("4x(MFMA + 16 fma)", t, out, 64, 4);
    run<5>("4x(MFMA + 24 fma)", t, out, 96, 4);
    run<6>("64 fma", t, out, 64, 0);
    run<7>("2x(MFMA C=0, MFMA acc)", t, out, 0, 4);
    run<8>("2x(pair, then 32 fma)", t, out, 64, 4);
    run<9>("2x(MFMA,16 fma,MFMA,16 fma)", t, out, 64, 4);
  }
  return 0;
}
