// Probe (round 6): beside the block-scaled BF6 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, cbsz = blgp = 3), is the de-quantisation cheaper
// as packed FP32 (v_pk_mul_f32 / v_pk_fma_f32: two lanes' worth per instruction) than as scalar v_mul_f32 / v_fma_f32?
// Per MFMA: 6 scalar (2 mul + 4 fma: the shipped kernel with pair-shared scales) against 3 packed (1 pk_mul + 2 pk_fma).
// 1, 2 and 4 waves per SIMD, all 256 CUs busy; s_memtime ticks = shader cycles.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/pk_mfma_probe.cpp -o build/pk_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define MA "v_mfma_scale_f32_16x16x128_f8f6f4 %0, %8, %9, 0, %10, %10 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n"
#define MB "v_mfma_scale_f32_16x16x128_f8f6f4 %1, %8, %9, 0, %10, %10 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n"
// scalar de-quantisation of accumulator X (4 floats) into c (4 floats): 2 products + 4 fma
#define SC6(c0, c1, c2, c3) "v_mul_f32 %6, %11, %12\n v_mul_f32 %7, %11, %13\n" \
  "v_fma_f32 " c0 ", %6, %12, " c0 "\n v_fma_f32 " c1 ", %6, %13, " c1 "\n v_fma_f32 " c2 ", %7, %12, " c2 "\n v_fma_f32 " c3 ", %7, %13, " c3 "\n"
// packed: 1 pk_mul + 2 pk_fma on register pairs
#define PK3(p0, p1) "v_pk_mul_f32 %6, %14, %15\n v_pk_fma_f32 " p0 ", %6, %15, " p0 "\n v_pk_fma_f32 " p1 ", %6, %14, " p1 "\n"
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, unsigned long long *cyc) {
  v6i fa = {0x0c30c30c, 0x30c30c30, (int)0xc30c30c3, 0x0c30c30c, 0x30c30c30, (int)threadIdx.x}, fb = fa;
  int sc = 127;
  v4f acc = {0, 0, 0, 0}, acc2 = {1, 1, 1, 1};
  float c[4] = {0.f, 1.f, 2.f, 3.f};
  v2f cp0 = {0.f, 1.f}, cp1 = {2.f, 3.f};
  float t0r = 0.f, t1r = 0.f;
  v2f tp = {0.f, 0.f};
  float s0 = 1.0001f, s1 = 0.9999f, s2 = 1.0002f;
  v2f sp0 = {1.0001f, 0.9999f}, sp1 = {1.0002f, 1.0003f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0)
      asm volatile(MA MB MA MB : "+v"(acc), "+v"(acc2), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(t0r), "+v"(t1r)
                   : "v"(fa), "v"(fb), "v"(sc), "v"(s0), "v"(s1), "v"(s2), "v"(sp0), "v"(sp1));
    if (MODE == 1)
      asm volatile(MA SC6("%2", "%3", "%4", "%5") MB SC6("%2", "%3", "%4", "%5") MA SC6("%2", "%3", "%4", "%5") MB SC6("%2", "%3", "%4", "%5")
                   : "+v"(acc), "+v"(acc2), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(t0r), "+v"(t1r)
                   : "v"(fa), "v"(fb), "v"(sc), "v"(s0), "v"(s1), "v"(s2), "v"(sp0), "v"(sp1));
    if (MODE == 2)
      asm volatile(MA PK3("%2", "%3") MB PK3("%2", "%3") MA PK3("%2", "%3") MB PK3("%2", "%3")
                   : "+v"(acc), "+v"(acc2), "+v"(cp0), "+v"(cp1), "+v"(c[2]), "+v"(c[3]), "+v"(tp), "+v"(t1r)
                   : "v"(fa), "v"(fb), "v"(sc), "v"(s0), "v"(s1), "v"(s2), "v"(sp0), "v"(sp1));
    if (MODE == 3)
      asm volatile(SC6("%2", "%3", "%4", "%5") SC6("%2", "%3", "%4", "%5") SC6("%2", "%3", "%4", "%5") SC6("%2", "%3", "%4", "%5")
                   : "+v"(acc), "+v"(acc2), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(t0r), "+v"(t1r)
                   : "v"(fa), "v"(fb), "v"(sc), "v"(s0), "v"(s1), "v"(s2), "v"(sp0), "v"(sp1));
    if (MODE == 4)
      asm volatile(PK3("%2", "%3") PK3("%2", "%3") PK3("%2", "%3") PK3("%2", "%3")
                   : "+v"(acc), "+v"(acc2), "+v"(cp0), "+v"(cp1), "+v"(c[2]), "+v"(c[3]), "+v"(tp), "+v"(t1r)
                   : "v"(fa), "v"(fb), "v"(sc), "v"(s0), "v"(s1), "v"(s2), "v"(sp0), "v"(sp1));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 7 && threadIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc2[1] + c[0] + c[1] + c[2] + c[3] + cp0.x + cp0.y + cp1.x + cp1.y + t0r + t1r + tp.x + tp.y;
}
template <int MODE> void run(const char *name, float *out) {
  static unsigned long long *cyc = nullptr; if (!cyc) (void)hipMalloc(&cyc, 8);
  printf("%-46s", name);
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, 200, cyc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, iters, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc; (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("  %dw/SIMD: %6.1f cyc (%6.1f ns) per wave and group", wps, (double)hc / iters / 4.0, ms * 1e6 / iters / 4.0);
  }
  printf("\n");
}
int main() {
  float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
  run<0>("MFMA alone", out);
  run<1>("MFMA + 2 v_mul + 4 v_fma (shipped)", out);
  run<2>("MFMA + 1 v_pk_mul + 2 v_pk_fma", out);
  run<3>("2 v_mul + 4 v_fma, no MFMA", out);
  run<4>("1 v_pk_mul + 2 v_pk_fma, no MFMA", out);
  return 0;
}
