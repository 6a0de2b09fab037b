// Probe (round 5): what one VALU instruction of the quantisers costs a SIMD, and what v_dot2_f32_f16 computes.
//   (a) issue cost in core cycles per instruction and SIMD with 1, 2 and 4 resident waves per SIMD (independent destinations, no waits);
//   (b) v_dot2_f32_f16 d = a.x * b.x + a.y * b.y + c against three candidate definitions (random halves incl. subnormals):
//       H1 fma(a.y, b.y, fma(a.x, b.x, c))   H2 RN(exact sum)   H3 RN(RN(a.x b.x + a.y b.y) + c)
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/rate_probe.cpp -o build/tools/rate_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int KIND>
__global__ __launch_bounds__(1024) void rate(float *out, int iters, unsigned long long *cyc) {
  float d[8];
  v2f p[8];
  for (int i = 0; i < 8; ++i) { d[i] = threadIdx.x * 0.25f + i; p[i] = v2f{d[i], d[i] + 1.f}; }
  float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  unsigned ah = 0x3C003C01u + threadIdx.x, bh = 0x38003801u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define ONE(i) \
    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a), "v"(b)); \
    else if constexpr (KIND == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(d[i]) : "v"(ah), "v"(bh)); \
    else if constexpr (KIND == 2) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "+v"(d[i]) : "v"(ah), "v"(b)); \
    else if constexpr (KIND == 3) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(d[i]) : "v"(ah), "v"(bh)); \
    else if constexpr (KIND == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7])); \
    else if constexpr (KIND == 5) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(d[i]) : "v"(ah), "v"(bh)); \
    else if constexpr (KIND == 6) asm volatile("v_max3_f16 %0, %1, %2, %0" : "+v"(d[i]) : "v"(ah), "v"(bh)); \
    else if constexpr (KIND == 7) asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(d[i]) : "v"(a)); \
    else if constexpr (KIND == 8) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(d[i]) : "v"(ah)); \
    else if constexpr (KIND == 9) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d[i]) : "v"(ah), "v"(bh), "v"(a)); \
    else if constexpr (KIND == 10) asm volatile("v_rcp_f32 %0, %1" : "=v"(d[i]) : "v"(a)); \
    else if constexpr (KIND == 11) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(d[i]) : "v"(ah), "v"(bh)); \
    else if constexpr (KIND == 12) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "+v"(d[i]) : "v"(ah), "v"(b)); \
    else if constexpr (KIND == 13) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d[i]) : "v"(a), "v"(b)); \
    else if constexpr (KIND == 14) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d[i]) : "v"(a), "v"(b)); \
    else asm volatile("v_and_b32 %0, %1, %2" : "=v"(d[i]) : "v"(ah), "v"(bh));
    REP8(ONE) REP8(ONE)
#undef ONE
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 7 && threadIdx.x == 0) cyc[0] = t1 - t0;
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += d[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void dot2(const unsigned *a, const unsigned *b, const float *c, float *o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d = c[i];
  asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a[i]), "v"(b[i]));
  o[i] = d;
}

static float h2f(uint16_t h) {
  const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
  float v = e == 0 ? ldexpf((float)m, -24) : ldexpf((float)(m + 1024), e - 25);
  return s ? -v : v;
}

template <int KIND>
static int run_rate(const char *name, float *out, unsigned long long *cyc) {
  const int iters = 4000;
  printf("%-18s", name);
  for (int waves = 1; waves <= 4; waves *= 2) {
    hipLaunchKernelGGL(rate<KIND>, dim3(256), dim3(256 * waves), 0, 0, out, iters, cyc);
    hipLaunchKernelGGL(rate<KIND>, dim3(256), dim3(256 * waves), 0, 0, out, iters, cyc);
    CK(hipDeviceSynchronize());
    unsigned long long c;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    // the SIMD issued waves * 16 * iters instructions while this wave ran its own 16 * iters
    printf("  %d wave/SIMD: %6.2f cyc per instr and SIMD", waves, (double)c / (16.0 * iters * waves));
  }
  printf("\n");
  return 0;
}

int main() {
  float *out; unsigned long long *cyc;
  CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 8));
  printf("# issue cost (s_memtime ticks; 100 MHz reference if the ratios read ~0.05, core clock otherwise)\n");
  run_rate<0>("v_fma_f32", out, cyc); run_rate<14>("v_mul_f32", out, cyc); run_rate<15>("v_and_b32", out, cyc);
  run_rate<1>("v_fma_mix_f32", out, cyc); run_rate<2>("v_fma_mixlo_f16", out, cyc); run_rate<12>("v_fma_mixhi_f16", out, cyc);
  run_rate<3>("v_pk_mul_f16", out, cyc); run_rate<11>("v_pk_add_f16", out, cyc); run_rate<4>("v_pk_fma_f32", out, cyc);
  run_rate<5>("v_dot2_f32_f16", out, cyc); run_rate<6>("v_max3_f16", out, cyc); run_rate<7>("v_add_f32_dpp", out, cyc);
  run_rate<8>("v_cvt_f32_f16", out, cyc); run_rate<13>("v_cvt_pk_f16_f32", out, cyc); run_rate<9>("v_perm_b32", out, cyc);
  run_rate<10>("v_rcp_f32", out, cyc);

  // ---- what v_dot2_f32_f16 computes
  const int n = 1 << 22;
  std::vector<unsigned> a(n), b(n); std::vector<float> c(n), o(n);
  uint64_t st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  for (int i = 0; i < n; ++i) {
    auto half = [&]() { uint16_t h; do { h = (uint16_t)rnd(); } while (((h >> 10) & 31) == 31); if ((i & 3) == 3) h &= 0x83FF | (((h >> 10) & 7) << 10); return h; };
    a[i] = half() | ((unsigned)half() << 16);
    b[i] = (i & 1) ? a[i] : (half() | ((unsigned)half() << 16));      // odd cases: squares, the quantiser's use
    float cf = ldexpf((float)(int)(rnd() & 0xFFFFFF) - 8388608.f, (int)(rnd() % 40) - 44);
    if ((i & 7) == 0) cf = 0.f;
    c[i] = cf;
  }
  unsigned *da, *db; float *dc, *dout;
  CK(hipMalloc(&da, n * 4)); CK(hipMalloc(&db, n * 4)); CK(hipMalloc(&dc, n * 4)); CK(hipMalloc(&dout, n * 4));
  CK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(dot2, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
  CK(hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost));
  long bad[5] = {0, 0, 0, 0, 0}; int shown = 0;
  for (int i = 0; i < n; ++i) {
    const float a0 = h2f(a[i] & 0xFFFF), a1 = h2f(a[i] >> 16), b0 = h2f(b[i] & 0xFFFF), b1 = h2f(b[i] >> 16);
    const float h1 = fmaf(a1, b1, fmaf(a0, b0, c[i]));
    const float h1r = fmaf(a0, b0, fmaf(a1, b1, c[i]));
    const long double ex = (long double)a0 * b0 + (long double)a1 * b1 + (long double)c[i];
    const float h2 = (float)ex;                                          // (64-bit significand: exact sum of the three, one rounding)
    const float h3 = (float)((double)a0 * b0 + (double)a1 * b1) + c[i];
    const float h4 = (a0 * b0 + a1 * b1) + c[i];                         // products exact in FP32; two roundings
    const float r = o[i];
    auto ne = [&](float x) { return memcmp(&x, &r, 4) != 0 && !(x == 0.f && r == 0.f); };
    bad[0] += ne(h1); bad[1] += ne(h1r); bad[2] += ne(h2); bad[3] += ne(h3); bad[4] += ne(h4);
    if (ne(h2) && shown < 6) { printf("  e.g. a=(%a,%a) b=(%a,%a) c=%a -> hw %a, one rounding %a, fma chain %a\n", a0, a1, b0, b1, c[i], r, h2, h1); ++shown; }
  }
  printf("# v_dot2_f32_f16 vs candidates over %d cases: mismatches  fma(a1 b1, fma(a0 b0, c)) %ld | fma(a0 b0, fma(a1 b1, c)) %ld | one rounding of the exact sum %ld | "
         "RN(RN(a0 b0 + a1 b1) + c) %ld | same in FP32 %ld\n", n, bad[0], bad[1], bad[2], bad[3], bad[4]);
  return 0;
}
