// Probes for the activation-quant kernels (gfx950):
//  (1) is v_cvt_rpi_i32_f32(|t|) == floor(|t| + 0.5) computed exactly (no double rounding at pred(0.5) etc.)
//      for every fp32 t in [0, 300)?  -> round-half-away in 3 instructions instead of 6
//  (2) is r1 = fma(fma(-r0, s, 1), r0, r0), r0 = v_rcp_f32(s) the correctly rounded 1/s for every positive fp16 s?
//      -> the exact 3-op quotient (div_probe.cpp) without the ~10-instruction IEEE division for r
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
__global__ void rpi_kernel(unsigned lo, unsigned n, unsigned long long *bad, unsigned *first) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float t = __uint_as_float(lo + i);
  int r;
  asm volatile("v_cvt_rpi_i32_f32 %0, |%1|" : "=v"(r) : "v"(t));
  const int want = (int)floor((double)t + 0.5);
  if (r != want) {
    if (atomicAdd(bad, 1ull) == 0) *first = lo + i;
  }
}
__global__ void rcp_kernel(unsigned long long *bad, unsigned *first) {
  const unsigned h = blockIdx.x * blockDim.x + threadIdx.x;      // fp16 bit pattern
  if (h == 0 || h >= 0x7C00) return;
  const _Float16 hs = __builtin_bit_cast(_Float16, (unsigned short)h);
  const float s = (float)hs;
  const float r0 = __builtin_amdgcn_rcpf(s);
  const float r1 = __builtin_fmaf(__builtin_fmaf(-r0, s, 1.0f), r0, r0);
  const float want = 1.0f / s;
  if (r1 != want) {
    if (atomicAdd(bad, 1ull) == 0) *first = h;
  }
}
int main() {
  unsigned long long *bad; unsigned *first;
  hipMalloc(&bad, 8); hipMalloc(&first, 4);
  hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
  const unsigned hi = 0x43960000u;     // 300.0f
  rpi_kernel<<<(hi + 255) / 256, 256>>>(0u, hi, bad, first);
  unsigned long long b; unsigned f;
  hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
  float ff; std::memcpy(&ff, &f, 4);
  printf("PROBE rpi: %llu mismatches of %u (first 0x%08x = %.9g)\n", b, hi, f, ff);
  hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
  rcp_kernel<<<0x7C00 / 256, 256>>>(bad, first);
  hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
  printf("PROBE rcp+newton: %llu mismatches of 31743 positive fp16 (first 0x%04x)\n", b, f);
  return 0;
}
