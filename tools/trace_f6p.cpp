// Dump the s_memtime stamps of workgroup 0 of the pipelined F6 kernel (gemm_w4a4_f6p_kernel, tools build of the library:
// make -C atom_amd/csrc tools; ATOM_F6_CFG=1016).  build/tools/trace_f6p [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../include/atom_hip.h"
int main(int argc, char **argv) {
  int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  int K4 = K - 128, G = K4 / 128;
  std::mt19937_64 rng(1);
  auto sc = [&]() { return (_Float16)(0.005f + 0.045f * ((rng() >> 11) * (1.0 / 9007199254740992.0))); };
  auto mk = [&](size_t bytes, int kind) { void *d; (void)hipMalloc(&d, bytes); std::vector<uint8_t> h(bytes);
    if (kind == 0) for (auto &x : h) x = rng() & 0xFF;
    else if (kind == 1) { _Float16 *p = (_Float16 *)h.data(); for (size_t i = 0; i < bytes / 2; ++i) p[i] = sc(); }
    else for (size_t r = 0; r < bytes / 104; ++r) { for (int i = 0; i < 96; ++i) h[r * 104 + i] = rng() & 0xFF; const _Float16 v = sc(); *(_Float16 *)&h[r * 104 + 96] = v; *(float *)&h[r * 104 + 100] = (float)v; }
    (void)hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice); return d; };
  const size_t Mp = (M + 255) / 256 * 256, Np = (N + 255) / 256 * 256;
  void *A4 = mk((size_t)G * Mp * 104, 2), *B4 = mk((size_t)G * Np * 104, 2), *A8 = mk((size_t)M * 128, 0), *B8 = mk((size_t)N * 128, 0);
  void *sA = mk((size_t)G * M * 2, 1), *sB = mk((size_t)G * N * 2, 1), *sA8 = mk(M * 2, 1), *sB8 = mk(N * 2, 1);
  void *D; (void)hipMalloc(&D, (size_t)M * N * 2);
  const size_t TR = 8 * 80;
  unsigned *tr; (void)hipMalloc(&tr, TR * 4); (void)hipMemset(tr, 0, TR * 4);
  char buf[64]; snprintf(buf, sizeof buf, "%llx", (unsigned long long)tr); setenv("ATOM_TRACE_PTR", buf, 1); setenv("ATOM_F6_CFG", "1016", 1);
  for (int i = 0; i < 200; ++i) { int st = atom_gemm_w4a4_f16(A4, B4, sA, sB, A8, B8, sA8, sB8, D, M, N, K, 128, 128, ATOM_AB_F6 | ATOM_SCALE_LAYOUT_PLAIN, nullptr); if (st) { printf("err %d\n", st); return 1; } }
  (void)hipDeviceSynchronize();
  std::vector<unsigned> h(TR); (void)hipMemcpy(h.data(), tr, TR * 4, hipMemcpyDeviceToHost);
  const unsigned t00 = h[0];
  auto d = [&](unsigned a, unsigned b) { return (int)(a - b); };
  for (int w = 0; w < 8; ++w) {
    unsigned *e = &h[w * 80];
    const double mhz = 100.0 * d(e[9], e[0]) / (double)d(e[79], e[78]);
    printf("wave %d: entry %d | dma issued +%d | stages 0,1 landed +%d | first fragments +%d | int4 loop +%d | keeper0 +%d | barrier +%d | keeper1 +%d | "
           "stores issued +%d | stores done +%d | total %d cycles, %.0f MHz\n", w, d(e[0], t00), d(e[1], e[0]), d(e[2], e[1]), d(e[3], e[2]), d(e[4], e[3]),
           d(e[5], e[4]), d(e[6], e[5]), d(e[7], e[6]), d(e[8], e[7]), d(e[9], e[8]), d(e[9], e[0]), mhz);
    for (int j = 0; j < 11; ++j) {
      unsigned *q = e + 10 + 6 * j;
      if (!q[0] && !q[5]) continue;
      printf("  traced step %2d: t=%7d  slots0-3 %5d  4-7 %5d  sync %5d  8-11 %5d  12-15 %5d  total %5d\n", j, d(q[0], e[0]), d(q[1], q[0]),
             d(q[2], q[1]), d(q[3], q[2]), d(q[4], q[3]), d(q[5], q[4]), d(q[5], q[0]));
    }
  }
  return 0;
}
