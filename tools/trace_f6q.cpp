// Dump the s_memtime stamps of the first and the last workgroup of the F6 headline kernel (gemm_w4a4_f6q_kernel<..., TR = true>, tools build
// of the library: make -C atom_amd/csrc tools; ATOM_F6_CFG=2016).  build/tools/trace_f6q [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include "../include/atom_hip.h"
int main(int argc, char **argv) {
  int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  const bool qk = argc > 4 && !strcmp(argv[4], "qk");       // the two-K-group kernel (ATOM_F6_CFG=2017)
  int K4 = K - 128, G = K4 / 128;
  std::mt19937_64 rng(1);
  auto sc = [&]() { return (_Float16)(0.005f + 0.045f * ((rng() >> 11) * (1.0 / 9007199254740992.0))); };
  auto mk = [&](size_t bytes, int kind) { void *d; (void)hipMalloc(&d, bytes); std::vector<uint8_t> h(bytes);
    if (kind == 0) for (auto &x : h) x = rng() & 0xFF;
    else if (kind == 1) { _Float16 *p = (_Float16 *)h.data(); for (size_t i = 0; i < bytes / 2; ++i) p[i] = sc(); }
    (void)hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice); return d; };
  const size_t Mp = (M + 255) / 256 * 256;
  // activations: random 6-bit fields are not all valid BF6 codes of integers, but the kernel's timing does not care; scales in the rows
  void *A6; { const size_t bytes = (size_t)G * Mp * 104; (void)hipMalloc(&A6, bytes); std::vector<uint8_t> h(bytes);
    for (size_t r = 0; r < bytes / 104; ++r) { for (int i = 0; i < 96; ++i) h[r * 104 + i] = rng() & 0xFF; const _Float16 v = sc(); *(_Float16 *)&h[r * 104 + 96] = v; *(float *)&h[r * 104 + 100] = (float)v; }
    (void)hipMemcpy(A6, h.data(), bytes, hipMemcpyHostToDevice); }
  void *B4 = mk((size_t)N * K4 / 2, 0), *A8 = mk((size_t)M * 128, 0), *B8 = mk((size_t)N * 128, 0);
  void *sA = mk((size_t)G * M * 2, 1), *sB = mk((size_t)G * N * 2, 1), *sA8 = mk(M * 2, 1), *sB8 = mk(N * 2, 1);
  void *B6; (void)hipMalloc(&B6, atom_f6_weight_bytes(N, K));
  if (int st = atom_repack_weight_f6s(B4, sB, N, K, B6, nullptr)) { printf("repack err %d\n", st); return 1; }
  void *D; (void)hipMalloc(&D, (size_t)M * N * 2);
  const size_t WS = qk ? 64 : 128;                          // dwords per wave (the q kernel: + the in-step stamps of K step 9)
  const size_t TR = 2 * 8 * WS;
  unsigned *tr; (void)hipMalloc(&tr, TR * 4); (void)hipMemset(tr, 0, TR * 4);
  char buf[64]; snprintf(buf, sizeof buf, "%llx", (unsigned long long)tr); setenv("ATOM_TRACE_PTR", buf, 1); setenv("ATOM_F6_CFG", qk ? "2017" : "2016", 1);
  const int layout = ATOM_AB_F6 | ATOM_B_F6S | ATOM_SCALE_LAYOUT_PLAIN;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 300; ++i) { int st = atom_gemm_w4a4_f16(A6, B6, sA, sB, A8, B8, sA8, sB8, D, M, N, K, 128, 128, layout, nullptr); if (st) { printf("err %d\n", st); return 1; } }
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < 200; ++i) (void)atom_gemm_w4a4_f16(A6, B6, sA, sB, A8, B8, sA8, sB8, D, M, N, K, 128, 128, layout, nullptr);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("traced build: %.2f us per launch (200 launches)\n", ms * 1e3 / 200);
  std::vector<unsigned> h(TR); (void)hipMemcpy(h.data(), tr, TR * 4, hipMemcpyDeviceToHost);
  auto d = [&](unsigned a, unsigned b) { return (int)(a - b); };
  if (qk) {
    for (int wg = 0; wg < 2; ++wg) {
      const unsigned t00 = h[wg * 8 * 64];
      printf("workgroup %s\n", wg ? "last" : "0");
      for (int w = 0; w < 8; ++w) {
        unsigned *e = &h[(wg * 8 + w) * 64];
        printf(" group %d wave %d: entry %+d | dma issued +%d | stages 0,1 landed +%d | first fragments +%d | int4 loop +%d | keeper barrier +%d | keeper +%d | "
               "catch-up + ring barrier +%d | exchange written + barrier +%d | sums + stores issued +%d | stores done +%d | total %d cycles = %.2f us, %.0f MHz\n",
               w / 4, w % 4, d(e[0], t00), d(e[1], e[0]), d(e[2], e[1]), d(e[3], e[2]), d(e[4], e[3]), w < 4 ? 0 : d(e[5], e[4]), w < 4 ? d(e[6], e[4]) : d(e[6], e[5]),
               d(e[7], e[6]), d(e[8], e[7]), d(e[9], e[8]), d(e[10], e[9]), d(e[10], e[0]), d(e[63], e[62]) / 100.0, 100.0 * d(e[10], e[0]) / (double)d(e[63], e[62]));
      }
      for (int w = 0; w < 8; w += 4) {
        unsigned *e = &h[(wg * 8 + w) * 64];
        const int n4 = w == 0 ? (G + 1) / 2 : G - (G + 1) / 2;
        printf(" group %d wave 0, K steps (cycles):", w / 4);
        for (int s2 = 0; s2 + 1 < n4 && s2 < 43; ++s2) printf(" %d", d(e[16 + s2 + 1], e[16 + s2]));
        printf(" | last %d\n", d(e[4], e[16 + n4 - 1]));
      }
    }
    return 0;
  }
  for (int wg = 0; wg < 2; ++wg) {
    const unsigned t00 = h[wg * 8 * WS];
    printf("workgroup %s\n", wg ? "last" : "0");
    for (int w = 0; w < 8; ++w) {
      unsigned *e = &h[(wg * 8 + w) * WS];
      const double mhz = 100.0 * d(e[8], e[0]) / (double)d(e[63], e[62]);
      printf(" wave %d: entry %+d | dma issued +%d | stages 0,1 landed +%d | first fragments +%d | int4 loop +%d | keeper0 +%d | barrier +%d | keeper1 + stores issued +%d | "
             "stores done +%d | total %d cycles = %.2f us, %.0f MHz\n", w, d(e[0], t00), d(e[1], e[0]), d(e[2], e[1]), d(e[3], e[2]), d(e[4], e[3]),
             d(e[5], e[4]), d(e[6], e[5]), d(e[7], e[6]), d(e[8], e[7]), d(e[8], e[0]), d(e[63], e[62]) / 100.0, mhz);
    }
    unsigned *e = &h[wg * 8 * WS];
    printf(" wave 0, K steps (cycles):");
    for (int s2 = 0; s2 + 1 < G && s2 < 43; ++s2) printf(" %d", d(e[16 + s2 + 1], e[16 + s2]));
    printf("\n wave 4, K steps (cycles):");
    e = &h[(wg * 8 + 4) * WS];
    for (int s2 = 0; s2 + 1 < G && s2 < 43; ++s2) printf(" %d", d(e[16 + s2 + 1], e[16 + s2]));
    printf("\n");
    // inside K step 9 (stamped: the stamps and the lgkmcnt(0) in front of each slot's second one lengthen this step): per pair slot
    // [MFMA pair issue | loads + DMA issue + read wait | de-quantisation of the previous pair], and the mid-step wait + barrier
    for (int w = 0; w < 8; w += 4) {
      unsigned *t = &h[(wg * 8 + w) * WS + 64];
      if (!t[0]) continue;
      int mf = 0, ld = 0, dq = 0;
      printf(" wave %d, K step 9 by pair slot: mfma | loads+dma+wait | dequant (cycles):", w);
      for (int i = 0; i < 16; ++i) {
        const int a = d(t[32 + 2 * i], i == 8 ? t[17] : t[i]), b = d(t[33 + 2 * i], t[32 + 2 * i]), c2 = d(i < 15 ? t[i + 1] : t[18], t[33 + 2 * i]);
        mf += a; ld += b; dq += c2;
        printf(" %d|%d|%d", a, b, c2);
      }
      printf("\n   = MFMA pair issue %d + fragment / scale reads, their wait and the LDS-DMA issue %d + de-quantisation %d + vmcnt wait and barrier %d "
             "= %d cycles (the step: %d)\n", mf, ld, dq, d(t[17], t[16]), mf + ld + dq + d(t[17], t[16]), d(t[18], t[0]));
    }
  }
  return 0;
}
