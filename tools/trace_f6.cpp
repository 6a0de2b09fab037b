// Dump s_memtime stamps of workgroup 0 of the F6 prefill kernel (tools build of the library: make -C atom_amd/csrc tools,
// ATOM_F6_CFG=116; first-generation 32x32x64 kernel).  build/tools/trace_f6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../include/atom_hip.h"
int main() {
  int M = 4096, N = 4096, K = 4096, K4 = K - 128, G = K4 / 128;
  std::mt19937_64 rng(1);
  auto sc = [&]() { return (_Float16)(0.005f + 0.045f * ((rng() >> 11) * (1.0 / 9007199254740992.0))); };
  auto mk = [&](size_t bytes, int kind) { void *d; (void)hipMalloc(&d, bytes); std::vector<uint8_t> h(bytes);
    if (kind == 0) for (auto &x : h) x = rng() & 0xFF;
    else if (kind == 1) { _Float16 *p = (_Float16 *)h.data(); for (size_t i = 0; i < bytes / 2; ++i) p[i] = sc(); }
    else for (size_t r = 0; r < bytes / 104; ++r) { for (int i = 0; i < 96; ++i) h[r * 104 + i] = rng() & 0xFF; *(_Float16 *)&h[r * 104 + 96] = sc(); }
    (void)hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice); return d; };
  void *A4 = mk((size_t)G * M * 104, 2), *B4 = mk((size_t)G * N * 104, 2), *A8 = mk((size_t)M * 128, 0), *B8 = mk((size_t)N * 128, 0);
  void *sA = mk((size_t)G * M * 2, 1), *sB = mk((size_t)G * N * 2, 1), *sA8 = mk(M * 2, 1), *sB8 = mk(N * 2, 1);
  void *D; (void)hipMalloc(&D, (size_t)M * N * 2);
  const size_t TR = 8 * 64 * 8;
  unsigned long long *tr; (void)hipMalloc(&tr, TR * 8); (void)hipMemset(tr, 0, TR * 8);
  char buf[64]; snprintf(buf, sizeof buf, "%llx", (unsigned long long)tr); setenv("ATOM_TRACE_PTR", buf, 1); setenv("ATOM_F6_CFG", "116", 1);
  for (int i = 0; i < 20; ++i) { int st = atom_gemm_w4a4_f16(A4, B4, sA, sB, A8, B8, sA8, sB8, D, M, N, K, 128, 128, ATOM_AB_F6, nullptr); if (st) { printf("err %d\n", st); return 1; } }
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(TR); (void)hipMemcpy(h.data(), tr, TR * 8, hipMemcpyDeviceToHost);
  const unsigned long long t00 = h[0];
  for (int w = 0; w < 8; ++w) {
    printf("wave %d: start %lld\n", w, (long long)(h[(w * 64) * 8] - t00));
    for (int s = 4; s < 10; ++s) {
      unsigned long long *e = &h[(w * 64 + s) * 8], *n = &h[(w * 64 + s + 1) * 8];
      printf("  step %2d: t=%7lld  vmcnt %4lld  barrier %5lld  frags %4lld  tiles01 %5lld  23 %5lld  45 %5lld  67 %5lld  tail %4lld  total %6lld\n", s,
             (long long)(e[0] - t00), (long long)(e[1] - e[0]), (long long)(e[2] - e[1]), (long long)(e[3] - e[2]), (long long)(e[4] - e[3]),
             (long long)(e[5] - e[4]), (long long)(e[6] - e[5]), (long long)(e[7] - e[6]), (long long)(n[0] - e[7]), (long long)(n[0] - e[0]));
    }
    printf("  last int4 step starts t=%lld\n", (long long)(h[(w * 64 + 30) * 8] - t00));
  }
  return 0;
}
