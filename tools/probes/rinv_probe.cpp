// Probe (round 5): rinv_sqrt_exact() of quant_math.h against the compiler's generic `1.0f / sqrtf(x)` (both operations correctly rounded)
// for every float in a set of binades around the RMSNorm statistic's range and a strided sweep of the whole fast range.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -I atom_amd/csrc -I include tools/probes/rinv_probe.cpp -o build/tools/rinv_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "common.h"
#include "quant_math.h"
__global__ void cmp(uint32_t first, uint32_t stride, uint64_t n, unsigned long long *bad, uint32_t *ex) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t u = first + (uint32_t)(i * stride);
  const float x = __uint_as_float(u);
  const float a = atom::rinv_sqrt_exact(x);
  volatile float xs = x;                                     // (keep the generic expansion: no common subexpression with the fast form)
  const float b = 1.0f / sqrtf(xs);
  if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) {
    if (atomicAdd(bad, 1ull) == 0) { ex[0] = u; ex[1] = __float_as_uint(a); ex[2] = __float_as_uint(b); }
  }
}
int main() {
  unsigned long long *bad; uint32_t *ex;
  hipMalloc(&bad, 8); hipMalloc(&ex, 12); hipMemset(bad, 0, 8);
  unsigned long long total = 0;
  auto run = [&](uint32_t first, uint32_t stride, uint64_t n, const char *what) {
    hipLaunchKernelGGL(cmp, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, first, stride, n, bad, ex);
    hipDeviceSynchronize();
    unsigned long long b; hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost);
    total += n;
    printf("%-58s %12llu arguments, mismatches so far %llu\n", what, (unsigned long long)n, b);
  };
  // every float of the binades 2^-40 .. 2^36 (exponent fields 87 .. 163): 77 x 2^23
  run(87u << 23, 1, 77ull << 23, "all floats in [2^-40, 2^37)");
  // the whole fast range [2^-64, 2^64] with stride 3, and its edges / the generic range around it
  run(63u << 23, 3, ((129ull << 23) + 2) / 3, "[2^-64, 2^65) stride 3");
  run((63u << 23) - 4096, 1, 8192, "around 2^-64 (fast / generic boundary)");
  run((191u << 23) - 4096, 1, 8192, "around 2^64");
  run(0, 1, 1 << 16, "zero and the smallest denormals");
  run(0x7F7F0000u, 1, 1 << 17, "largest floats, inf, NaNs");
  unsigned long long b; hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost);
  if (b) { uint32_t e[3]; hipMemcpy(e, ex, 12, hipMemcpyDeviceToHost); printf("first mismatch: x=%08x fast=%08x generic=%08x\n", e[0], e[1], e[2]); }
  printf("rinv_sqrt_exact vs 1.0f / sqrtf(x): %llu arguments, %llu mismatches\n", total, b);
  return b != 0;
}
