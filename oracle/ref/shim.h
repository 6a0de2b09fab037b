// Host shim for the CUDA vocabulary the reference's CPU golden functions use (oracle/ref/extract.py cuts them out of
// test_Reorder.cu, test_activate.cu, test_RMSNorm.cu).  Test infrastructure only.
//   half           cuda_fp16.h's type on the host: IEEE binary16 storage, float <-> half conversions round to nearest even
//                  (what __float2half / __half2float do); everything else the goldens do with it goes through float
//   HOST_DEVICE    plain inline
//   max / min      the CUDA math overloads the reference's clamp() relies on
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include <cstring>

namespace shim {
inline float h2f(uint16_t h) {                       // binary16 -> binary32, exact
  uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, u;
  if (e == 0) {
    if (m == 0) u = s;
    else { int sh = 0; while (!(m & 0x400u)) { m <<= 1; ++sh; } m &= 0x3FFu; u = s | ((uint32_t)(113 - sh) << 23) | (m << 13); }
  } else if (e == 31) u = s | 0x7F800000u | (m << 13);
  else u = s | ((e + 112) << 23) | (m << 13);
  float f; std::memcpy(&f, &u, 4); return f;
}
inline uint16_t f2h(float f) {                       // binary32 -> binary16, round to nearest even
  uint32_t u; std::memcpy(&u, &f, 4);
  uint32_t s = (u >> 16) & 0x8000u, a = u & 0x7FFFFFFFu;
  if (a >= 0x7F800000u) return (uint16_t)(s | 0x7C00u | (a > 0x7F800000u ? 0x200u : 0));
  if (a >= 0x477FF000u) return (uint16_t)(s | 0x7C00u);
  if (a < 0x33000001u) return (uint16_t)s;
  int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7FFFFFu) | 0x800000u;
  int shift = e < -14 ? (13 + (-14 - e)) : 13;
  uint32_t r = m >> shift, rem = m & ((1u << shift) - 1), hf = 1u << (shift - 1);
  if (rem > hf || (rem == hf && (r & 1))) ++r;
  if (e < -14) return (uint16_t)(s | r);
  return (uint16_t)(s + ((uint32_t)(e + 15) << 10) + (r - 0x400u));
}
}  // namespace shim

struct half {
  uint16_t bits;
  half() = default;
  half(float f) : bits(shim::f2h(f)) {}
  half(double f) : bits(shim::f2h((float)f)) {}
  operator float() const { return shim::h2f(bits); }
};
static_assert(sizeof(half) == 2, "half must be 2 bytes");

#define HOST_DEVICE inline
using std::max;
using std::min;

// ---- for the flashinfer CPU reference (kernels/src/flashinfer/cpu_reference.h and the definitions it uses)
#include <cassert>
#include <iostream>
#include <string>
#include <type_traits>
#include <vector>
struct half2 { half x, y; };
inline float __half2float(half h) { return (float)h; }
// ---- for the u4-output GEMM epilogue (DenseLayerGEMM_i4_o4.cu:72-80, :722-786)
struct float4 { float x, y, z, w; };
inline half2 __floats2half2_rn(float a, float b) { return half2{half(a), half(b)}; }
#define __host__
#define __device__
#define __forceinline__ inline
#define FLASHINFER_INLINE inline
