// Native bench of the fused activation-quant ops (HBM-bound): hipEvent timing, algorithmic GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include <numeric>
#include "../include/atom_hip.h"
int main(int argc, char **argv) {
  int M = argc > 1 ? atoi(argv[1]) : 4096, H = argc > 2 ? atoi(argv[2]) : 4096, iters = argc > 3 ? atoi(argv[3]) : 200;
  const bool noidx = argc > 4 && atoi(argv[4]) == 1;   // 1: identity order (no gather)
  const bool seqidx = argc > 4 && atoi(argv[4]) == 2;  // 2: identity permutation passed as an index
  std::mt19937 rng(1);
  std::vector<_Float16> hx((size_t)M * H), hw(H);
  std::normal_distribution<float> nd(0, 1);
  for (auto &v : hx) v = (_Float16)nd(rng);
  for (auto &v : hw) v = (_Float16)(1.0f + 0.1f * nd(rng));
  std::vector<int16_t> hidx(H); std::iota(hidx.begin(), hidx.end(), 0); if (!seqidx) std::shuffle(hidx.begin(), hidx.end(), rng);
  void *x, *b, *w, *idx, *o8, *o4, *s8, *s4, *xq;
  hipMalloc(&x, hx.size() * 2); hipMalloc(&b, hx.size() * 2); hipMalloc(&w, H * 2); hipMalloc(&idx, H * 2);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(b, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), H * 2, hipMemcpyHostToDevice); hipMemcpy(idx, hidx.data(), H * 2, hipMemcpyHostToDevice);
  hipMalloc(&o8, (size_t)M * 128); hipMalloc(&o4, (size_t)(H / 128) * ((size_t)(M + 255) / 256 * 256) * 104 + (size_t)M * H);  /* large enough for every code format */ hipMalloc(&s8, (size_t)M * 8 + 256); hipMalloc(&s4, (size_t)(H / 128) * (M * 8 + 256));
  hipMalloc(&xq, hx.size() * 2);
  const int fmt = getenv("ATOM_QB_FMT") ? atoi(getenv("ATOM_QB_FMT")) : 0;   // 0 packed, 0x100 wide, 0x200 F6
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double out_bytes = (double)M * ((H - 128) / 2 + 128 + 2.0 * (H / 128));
  for (int op = 0; op < 3; ++op) for (int mode = 0; mode < 2; ++mode) for (int dq = 0; dq < 2; ++dq) {
    auto run = [&]() {
      void *xo = dq ? xq : nullptr;
      if (op == 0) return atom_reorder_quant_f16(x, noidx ? nullptr : (const int16_t *)idx, M, H, mode | fmt, mode ? 0.9f : 1.0f, 1, o8, o4, s8, s4, xo, nullptr);
      if (op == 1) return atom_rmsnorm_reorder_quant_f16(x, w, 1e-5f, (const int16_t *)idx, M, H, mode | fmt, mode ? 0.9f : 1.0f, 1, o8, o4, s8, s4, xo, nullptr);
      return atom_silu_mul_quant_f16(x, b, M, H, mode | fmt, mode ? 0.9f : 1.0f, 1, o8, o4, s8, s4, xo, nullptr);
    };
    for (int i = 0; i < 10; ++i) { int st = run(); if (st) { printf("error %d (%s) op=%d mode=%d dq=%d\n", st, atom_strerror(st), op, mode, dq); return 1; } }
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) run();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    const double in_bytes = (double)M * H * 2 * (op == 2 ? 2 : 1);
    const double bytes = in_bytes + out_bytes + (dq ? (double)M * H * 2 : 0);
    printf("QRESULT op=%s mode=%s dequant_out=%d M=%d H=%d  %.2f us  %.0f GB/s algorithmic (%.1f MB)\n",
           op == 0 ? "reorder" : op == 1 ? "rmsnorm" : "silu_mul", mode ? "sim" : "kernel", dq, M, H, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes / 1e6);
  }
  return 0;
}
