// Dump s_memtime stamps of workgroup 0 of the prefill kernel (variant 310/311).  build/trace_gemm [variant]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../include/atom_hip.h"
int main(int argc, char **argv) {
  const char *variant = argc > 1 ? argv[1] : "310";
  int M = 4096, N = 4096, K = 4096, K4 = K - 128, G = K4 / 128;
  std::mt19937_64 rng(1);
  auto mk = [&](size_t bytes, bool scale) { void *d; hipMalloc(&d, bytes); std::vector<uint8_t> h(bytes);
    if (!scale) for (auto &x : h) x = rng() & 0xFF; else { _Float16 *p = (_Float16 *)h.data(); for (size_t i = 0; i < bytes / 2; ++i) p[i] = (_Float16)(0.005f + 0.045f * ((rng() >> 11) * (1.0 / 9007199254740992.0))); }
    hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice); return d; };
  const bool wide = atoi(variant) == 330;
  void *A4 = mk((size_t)M * K4 / (wide ? 1 : 2), false), *B4 = mk((size_t)N * K4 / 2, false), *A8 = mk((size_t)M * 128, false), *B8 = mk((size_t)N * 128, false);
  void *sA = mk((size_t)G * M * 2, true), *sB = mk((size_t)G * N * 2, true), *sA8 = mk(M * 2, true), *sB8 = mk(N * 2, true);
  void *D; hipMalloc(&D, (size_t)M * N * 2);
  unsigned long long *tr; hipMalloc(&tr, 8 * 64 * 4 * 8); hipMemset(tr, 0, 8 * 64 * 4 * 8);
  char buf[64]; snprintf(buf, sizeof buf, "%llx", (unsigned long long)tr); setenv("ATOM_TRACE_PTR", buf, 1); setenv("ATOM_GEMM_VARIANT", variant, 1);
  for (int i = 0; i < 20; ++i) { int st = atom_gemm_w4a4_f16(A4, B4, sA, sB, A8, B8, sA8, sB8, D, M, N, K, 128, 128, 1 | (wide ? ATOM_A_WIDE : 0), nullptr); if (st) { printf("err %d\n", st); return 1; } }
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(8 * 64 * 4); hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
  int nw = atoi(variant) == 311 ? 4 : 8;
  for (int w = 0; w < nw; ++w) {
    unsigned long long t00 = h[(0 * 64 + 0) * 4];
    printf("wave %d: start %lld\n", w, (long long)(h[(w * 64) * 4] - t00));
    for (int s = 4; s < 12; ++s) {
      unsigned long long *e = &h[(w * 64 + s) * 4], *n = &h[(w * 64 + s + 1) * 4];
      printf("  step %2d: t=%7lld  vmcnt-wait %5lld  barrier %5lld  dma-issue %5lld  compute %6lld  total %6lld\n", s, (long long)(e[0] - t00),
             (long long)(e[1] - e[0]), (long long)(e[2] - e[1]), (long long)(e[3] - e[2]), (long long)(n[0] - e[3]), (long long)(n[0] - e[0]));
    }
    printf("  loop end t=%lld (steps 0..33)\n", (long long)(h[(w * 64 + 33) * 4] - t00));
  }
  return 0;
}
