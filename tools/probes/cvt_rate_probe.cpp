// Probe (round 6): issue cost of the instructions a 4-bit -> FP32 widening + FMA could be built from, on gfx950, per SIMD, with 1 / 2 / 3
// waves per SIMD and all 256 CUs busy.  16 independent destination registers per stream, 64 instructions per loop trip.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cvt_rate_probe.cpp -o build/cvt_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define R16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define R64(OP) R16(OP) R16(OP) R16(OP) R16(OP)
#define OUTS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), \
               "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "v"(c), "v"(x)
#define OUTS2 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), "+v"(d[9]), \
                "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15]) : "v"(b2), "v"(c2), "v"(x)
#define I_FMA(n) "v_fma_f32 %" #n ", %16, %17, %" #n "\n"
#define I_MIX(n) "v_fma_mix_f32 %" #n ", %18, %17, %" #n " op_sel_hi:[1,0,0]\n"
#define I_MIXH(n) "v_fma_mix_f32 %" #n ", %18, %17, %" #n " op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define I_CVTH(n) "v_cvt_f32_f16_e32 %" #n ", %18\n"
#define I_UB0(n) "v_cvt_f32_ubyte0_e32 %" #n ", %18\n"
#define I_UB3(n) "v_cvt_f32_ubyte3_e32 %" #n ", %18\n"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %18, %16, %17\n"
#define I_U32(n) "v_cvt_f32_u32_e32 %" #n ", %18\n"
#define I_OFFI4(n) "v_cvt_off_f32_i4_e32 %" #n ", %18\n"
#define I_OFFI4S(n) "v_cvt_off_f32_i4_sdwa %" #n ", %18 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n"
#define I_DOT2(n) "v_dot2c_f32_f16_e32 %" #n ", %18, %17\n"
#define I_BFE(n) "v_bfe_u32 %" #n ", %18, 4, 4\n"
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %16, %17, %" #n "\n"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %16, %17\n"
#define I_PKADD(n) "v_pk_add_f32 %" #n ", %16, %" #n "\n"
#define I_PKFP8(n) "v_cvt_pk_f32_fp8_e32 %" #n ", %18\n"
#define I_PKFP8H(n) "v_cvt_pk_f32_fp8_sdwa %" #n ", %18 src0_sel:WORD_1\n"
#define I_PKBF8(n) "v_cvt_pk_f32_bf8_e32 %" #n ", %18\n"
#define I_F32FP8(n) "v_cvt_f32_fp8_e32 %" #n ", %18\n"
#define I_F32FP8B(n) "v_cvt_f32_fp8_sdwa %" #n ", %18 src0_sel:BYTE_3\n"
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, unsigned long long *cyc) {
  float a[16]; v2f d[16];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x + i; d[i] = v2f{a[i], a[i] + 1}; }
  float b = 1.0001f, c = 0.5f; v2f b2 = {1.0001f, 0.9f}, c2 = {0.5f, 0.25f};
  unsigned x = 0x3c003c00u + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) asm volatile(R64(I_FMA) OUTS);
    if (MODE == 1) asm volatile(R64(I_MIX) OUTS);
    if (MODE == 2) asm volatile(R64(I_MIXH) OUTS);
    if (MODE == 3) asm volatile(R64(I_CVTH) OUTS);
    if (MODE == 4) asm volatile(R64(I_UB0) OUTS);
    if (MODE == 5) asm volatile(R64(I_UB3) OUTS);
    if (MODE == 6) asm volatile(R64(I_ANDOR) OUTS);
    if (MODE == 7) asm volatile(R64(I_U32) OUTS);
    if (MODE == 8) asm volatile(R64(I_OFFI4) OUTS);
    if (MODE == 9) asm volatile(R64(I_OFFI4S) OUTS);
    if (MODE == 10) asm volatile(R64(I_DOT2) OUTS);
    if (MODE == 11) asm volatile(R64(I_BFE) OUTS);
    if (MODE == 12) asm volatile(R64(I_PKFMA) OUTS2);
    if (MODE == 13) asm volatile(R64(I_PKMUL) OUTS2);
    if (MODE == 14) asm volatile(R64(I_PKADD) OUTS2);
    if (MODE == 15) asm volatile(R64(I_PKFP8) OUTS2);
    if (MODE == 16) asm volatile(R64(I_PKFP8H) OUTS2);
    if (MODE == 17) asm volatile(R64(I_PKBF8) OUTS2);
    if (MODE == 18) asm volatile(R64(I_F32FP8) OUTS);
    if (MODE == 19) asm volatile(R64(I_F32FP8B) OUTS);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 7 && threadIdx.x == 0) cyc[0] = t1 - t0;
  float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + d[i].x + d[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// Two views per occupancy: "t0" = s_memtime ticks per instruction of wave 0 of one workgroup -- the OLDEST wave of its SIMD, served first:
// its pace barely changes with the waves beside it and says nothing about throughput --; "simd" = the whole launch by HIP events:
// nanoseconds per instruction and SIMD (what the hardware sustains).
template <int MODE> void run(const char *name, float *out) {
  static unsigned long long *cyc = nullptr; if (!cyc) (void)hipMalloc(&cyc, 8);
  printf("%-32s", name);
  for (int wps = 1; wps <= 4; ++wps) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, 2000, cyc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, iters, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc; (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("  %dw: t0 %5.2f simd %5.3f ns", wps, (double)hc / iters / 64.0, ms * 1e6 / iters / 64.0 / wps);
  }
  printf("\n");
}
int main() {
  float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
  run<0>("v_fma_f32", out);
  run<1>("v_fma_mix_f32 (f16 lo src0)", out);
  run<2>("v_fma_mix_f32 (f16 hi src0)", out);
  run<3>("v_cvt_f32_f16", out);
  run<4>("v_cvt_f32_ubyte0", out);
  run<5>("v_cvt_f32_ubyte3", out);
  run<6>("v_and_or_b32", out);
  run<7>("v_cvt_f32_u32", out);
  run<8>("v_cvt_off_f32_i4", out);
  run<9>("v_cvt_off_f32_i4 sdwa BYTE_2", out);
  run<10>("v_dot2c_f32_f16", out);
  run<11>("v_bfe_u32", out);
  run<12>("v_pk_fma_f32", out);
  run<13>("v_pk_mul_f32", out);
  run<14>("v_pk_add_f32", out);
  run<15>("v_cvt_pk_f32_fp8 (word 0)", out);
  run<16>("v_cvt_pk_f32_fp8 sdwa WORD_1", out);
  run<17>("v_cvt_pk_f32_bf8", out);
  run<18>("v_cvt_f32_fp8 (byte 0)", out);
  run<19>("v_cvt_f32_fp8 sdwa BYTE_3", out);
  return 0;
}
