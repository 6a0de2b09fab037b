// Dump the s_memtime stamps of workgroup 0 of the two-K-group F6 kernels (gemm_w4a4_f6x16_kernel<.., KG = 2>, tools build of the
// library: make -C atom_amd/csrc tools; ATOM_F6_CFG = 1005: 128x128 tile, 1009: 64x128).  build/tools/trace_f6kg cfg [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../include/atom_hip.h"
int main(int argc, char **argv) {
  const char *cfg = argc > 1 ? argv[1] : "1005";
  int M = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 4096, K = argc > 4 ? atoi(argv[4]) : 4096;
  int K4 = K - 128, G = K4 / 128;
  std::mt19937_64 rng(1);
  auto sc = [&]() { return (_Float16)(0.005f + 0.045f * ((rng() >> 11) * (1.0 / 9007199254740992.0))); };
  auto mk = [&](size_t bytes, int kind, size_t tail_floats = 0) { void *d; (void)hipMalloc(&d, bytes + tail_floats * 4); std::vector<uint8_t> h(bytes + tail_floats * 4);
    if (kind == 0) for (auto &x : h) x = rng() & 0xFF;
    else if (kind == 1) { _Float16 *p = (_Float16 *)h.data(); for (size_t i = 0; i < bytes / 2; ++i) p[i] = sc(); }
    else for (size_t r = 0; r < bytes / 104; ++r) { for (int i = 0; i < 96; ++i) h[r * 104 + i] = rng() & 0xFF; const _Float16 v = sc(); *(_Float16 *)&h[r * 104 + 96] = v; *(float *)&h[r * 104 + 100] = (float)v; }
    for (size_t i = 0; i < tail_floats; ++i) ((float *)(h.data() + bytes))[i] = (float)sc();
    (void)hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice); return d; };
  const size_t Mp = (M + 255) / 256 * 256, Np = (N + 255) / 256 * 256;
  void *A4 = mk((size_t)G * Mp * 104, 2), *B4 = mk((size_t)G * Np * 104, 2, (size_t)G * Np), *A8 = mk((size_t)M * 128, 0), *B8 = mk((size_t)N * 128, 0);
  void *sA = mk((size_t)G * M * 2, 1), *sB = mk((size_t)G * N * 2, 1), *sA8 = mk(M * 2, 1), *sB8 = mk(N * 2, 1);
  void *D; (void)hipMalloc(&D, (size_t)M * N * 2);
  const size_t TR = 8 * 64;
  unsigned *tr; (void)hipMalloc(&tr, TR * 4); (void)hipMemset(tr, 0, TR * 4);
  char buf[64]; snprintf(buf, sizeof buf, "%llx", (unsigned long long)tr); setenv("ATOM_TRACE_PTR", buf, 1); setenv("ATOM_F6_CFG", cfg, 1);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) atom_gemm_w4a4_f16(A4, B4, sA, sB, A8, B8, sA8, sB8, D, M, N, K, 128, 128, ATOM_AB_F6 | ATOM_B_F6S | ATOM_SCALE_LAYOUT_PLAIN, nullptr);
  (void)hipEventRecord(e0, nullptr);
  for (int i = 0; i < 200; ++i) { int st = atom_gemm_w4a4_f16(A4, B4, sA, sB, A8, B8, sA8, sB8, D, M, N, K, 128, 128, ATOM_AB_F6 | ATOM_B_F6S | ATOM_SCALE_LAYOUT_PLAIN, nullptr); if (st) { printf("err %d\n", st); return 1; } }
  (void)hipEventRecord(e1, nullptr);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("cfg %s  %dx%dx%d: %.2f us per launch (traced build)\n", cfg, M, N, K, ms * 5.0);
  std::vector<unsigned> h(TR); (void)hipMemcpy(h.data(), tr, TR * 4, hipMemcpyDeviceToHost);
  auto d = [&](unsigned a, unsigned b) { return (int)(a - b); };
  const unsigned t00 = h[1];
  for (int w = 0; w < 8; ++w) {
    unsigned *e = &h[w * 64];
    printf("wave %d: first stages issued at %d | loop done +%d | end +%d   (s_memtime ticks)\n", w, d(e[1], t00), d(e[62], e[1]), d(e[63], e[62]));
    for (int j = 0; j < 10; ++j) {
      unsigned *q = e + 2 + 6 * j;
      if (!q[0] && !q[5]) continue;
      printf("  step %2d: t=%6d  dma wait %4d  barrier %4d  first MFMA %4d  half %4d  rest %4d  total %5d\n", j, d(q[0], e[1]), d(q[1], q[0]), d(q[2], q[1]),
             d(q[3], q[2]), d(q[4], q[3]), d(q[5], q[4]), d(q[5], q[0]));
    }
  }
  return 0;
}
