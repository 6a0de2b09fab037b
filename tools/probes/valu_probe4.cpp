// Probe 2: hand-interleaved MFMA / VALU streams (all inline asm, so the order is what is written).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define F(n) "v_fma_f32 %" #n ", %" #n ", %18, %19\n"
#define MX(n) "v_fma_mix_f32 %" #n ", %" #n ", %18, %19 op_sel_hi:[0,1,0]\n"
#define AN(n) "v_and_b32 %" #n ", 0xf0f0f0f0, %" #n "\n"
#define SH(n) "v_lshlrev_b32 %" #n ", 4, %" #n "\n"
#define V4(OP) OP(2) OP(3) OP(4) OP(5)
#define V8(OP) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9)
#define V12(OP) V8(OP) OP(10) OP(11) OP(12) OP(13)
#define V16(OP) V8(OP) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15) OP(16) OP(17)
#define MA "v_mfma_i32_32x32x32_i8 %0, %20, %21, %0\n"
#define MB "v_mfma_i32_32x32x32_i8 %1, %20, %21, %1\n"
#define BODY(VS) MA VS MB VS MA VS MB VS
#define OPS : "+v"(acc), "+v"(acc2), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
              "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "v"(c), "v"(fa), "v"(fb)
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, unsigned long long *cyc) {
  float a[16]; for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
  float b = 1.0001f, c = 0.5f;
  v4i fa = {1, 2, 3, (int)threadIdx.x}, fb = {5, 6, 7, 8};
  v16i acc = {0}, acc2 = {1};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) asm volatile(BODY("") OPS);
    if (MODE == 1) asm volatile(BODY(V4(F)) OPS);
    if (MODE == 2) asm volatile(BODY(V8(F)) OPS);
    if (MODE == 3) asm volatile(BODY(V12(F)) OPS);
    if (MODE == 4) asm volatile(BODY(V16(F)) OPS);
    if (MODE == 5) asm volatile(BODY(V8(MX)) OPS);
    if (MODE == 6) asm volatile(BODY(V8(AN)) OPS);
    if (MODE == 7) asm volatile(BODY(V8(SH)) OPS);
    if (MODE == 8) asm volatile(V16(F) V16(F) OPS);          // 32 fma, no MFMA
    if (MODE == 9) asm volatile(V16(MX) V16(MX) OPS);
    if (MODE == 10) asm volatile(V16(AN) V16(SH) OPS);
    if (MODE == 11) asm volatile(MA MB MA MB V16(F) V16(F) OPS);   // phase-separated: 4 MFMA then 32 fma
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
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 100, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ns = ms * 1e6 / iters; int wps = threads / 256;
  unsigned long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  double cpi = (double)hc / iters;
  printf("%-28s w/SIMD %d: %7.1f ns/iter %8.1f memtime-ticks/iter (%.3f ticks/ns) | per SIMD: %6.2f ns/MFMA  %5.2f ns/VALU\n", name, wps, ns, cpi, cpi / ns,
         mfma ? ns / (mfma * wps) : 0.0, valu ? ns / (valu * wps) : 0.0);
}
int main() {
  float *out; hipMalloc(&out, 256 * 512 * 4);
  for (int t = 256; t <= 512; t += 256) {
    run<0>("4 MFMA", t, out, 0, 4);
    run<1>("4x(MFMA+4 fma)", t, out, 16, 4);
    run<2>("4x(MFMA+8 fma)", t, out, 32, 4);
    run<3>("4x(MFMA+12 fma)", t, out, 48, 4);
    run<4>("4x(MFMA+16 fma)", t, out, 64, 4);
    run<5>("4x(MFMA+8 fma_mix)", t, out, 32, 4);
    run<6>("4x(MFMA+8 and)", t, out, 32, 4);
    run<7>("4x(MFMA+8 lshl)", t, out, 32, 4);
    run<8>("32 fma", t, out, 32, 0);
    run<9>("32 fma_mix", t, out, 32, 0);
    run<10>("16 and + 16 lshl", t, out, 32, 0);
    run<11>("4 MFMA then 32 fma", t, out, 32, 4);
  }
  return 0;
}
