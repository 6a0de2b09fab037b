// Exhaustive proof on hardware (round 4): the FP16-domain form of the simulated-path code (quant_math.h sim_codes4: two mixed-precision
// FMAs around a half estimate, clamp, + 1536) equals group_code<true> -- rint(clamp(half(v / s))) with the FP32 exact quotient -- for
// EVERY group maximum a group can have (every positive finite half), both code widths (INT4 groups, the INT8 keeper), a set of clip
// ratios down to the 1/64 limit of the fast path, and EVERY half v with |v| <= amax.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I atom_amd/csrc tools/probes/simh_probe.cpp -o build/tools/simh_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "quant_math.h"
using namespace atom;

__global__ void k(float clip, int keeper, unsigned long long *bad, unsigned long long *tested, unsigned *ex) {
  const unsigned ab = blockIdx.x + 1;                       // amax bits 1 .. 0x7BFF
  const float amax = (float)__builtin_bit_cast(half_t, (unsigned short)ab);
  const GroupScale gs = group_scale<true>(amax, keeper != 0, clip);
  const unsigned s2 = (unsigned)__builtin_bit_cast(unsigned short, (half_t)gs.s_store);
  const unsigned lo2 = keeper ? 0xD800D800u : 0xC800C800u, hi2 = keeper ? 0x57F057F0u : 0x47004700u;
  unsigned long long nb = 0, nt = 0;
  // magnitudes 0 .. ab, both signs: thread handles 8 values (4 pairs) per step
  for (unsigned base = threadIdx.x * 8; base <= ab; base += blockDim.x * 8) {
    for (int sign = 0; sign < 2; ++sign) {
      unsigned vb[8], z[4], t[4];
      for (int i = 0; i < 8; ++i) vb[i] = (base + i <= ab ? base + i : ab) | (sign << 15);
      for (int i = 0; i < 4; ++i) z[i] = vb[i] | (vb[i + 4] << 16);
      sim_codes4(z[0], z[1], z[2], z[3], gs.rs, s2, lo2, hi2, t[0], t[1], t[2], t[3]);
      for (int i = 0; i < 8; ++i) {
        const float v = (float)__builtin_bit_cast(half_t, (unsigned short)vb[i]);
        const int want = (int)group_code<true>(v, gs);
        const int got = (int)((t[i & 3] >> (16 * (i >> 2))) & 0xFFFF) - 0x6600;
        ++nt;
        if (want != got) {
          ++nb;
          if (atomicAdd(&ex[0], 1u) < 8) { unsigned j = atomicAdd(&ex[1], 1u); ex[2 + 4 * j] = ab; ex[3 + 4 * j] = vb[i]; ex[4 + 4 * j] = (unsigned)want; ex[5 + 4 * j] = (unsigned)got; }
        }
      }
    }
  }
  if (nb) atomicAdd(bad, nb);
  atomicAdd(tested, nt);
}

int main() {
  unsigned long long *bad, *tested; unsigned *ex;
  hipMalloc(&bad, 8); hipMalloc(&tested, 8); hipMalloc(&ex, 4 * 64);
  const float clips[] = {1.0f, 0.9f, 0.85f, 0.5f, 1.0f / 64.0f};
  unsigned long long total_bad = 0;
  for (int keeper = 0; keeper < 2; ++keeper)
    for (float clip : clips) {
      if (keeper && clip != 1.0f) continue;
      hipMemset(bad, 0, 8); hipMemset(tested, 0, 8); hipMemset(ex, 0, 4 * 64);
      hipLaunchKernelGGL(k, dim3(0x7BFF), dim3(256), 0, 0, clip, keeper, bad, tested, ex);
      unsigned long long h = 0, n = 0; unsigned hex[64];
      hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&n, tested, 8, hipMemcpyDeviceToHost); hipMemcpy(hex, ex, sizeof hex, hipMemcpyDeviceToHost);
      printf("SIMHPROBE keeper %d clip %.6f: (amax, v) pairs tested %llu, code mismatches %llu\n", keeper, clip, n, h);
      for (unsigned i = 0; i < hex[1] && i < 8; ++i)
        printf("  example amax=0x%04x v=0x%04x want %d got %d\n", hex[2 + 4 * i], hex[3 + 4 * i], (int)hex[4 + 4 * i], (int)hex[5 + 4 * i]);
      total_bad += h;
    }
  printf("SIMHPROBE total mismatches %llu\n", total_bad);
  return total_bad != 0;
}
