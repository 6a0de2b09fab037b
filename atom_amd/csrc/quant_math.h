// The per-group quantisation arithmetic of the fused activation ops, shared by quant_kernels.hip and the GEMM epilogue that fuses
// SiLU x up -> quant behind gate_proj / up_proj (gemm_w4a4_f6.hip).  Two modes, specified op by op and restated in oracle/:
//   SIM     the simulated path (model/quant.py:134-181): FP16 opmath -- amax.clamp(1e-5) * clip, scale = amax / qmax, every step
//           rounded to half; code = clamp(round_half_even(x / scale))
//   kernel  the CUDA kernels (Reorder.cuh:137-178 == Activate.cuh:112-167): FP32, scale = amax * clip / qmax,
//           code = clamp(round_half_away(x * (1 / scale)))
#pragma once
#include "common.h"

namespace atom {

struct GroupScale {
  float s_store;     // the scale as stored (SIM: already a half value; kernel: FP32, rounded to half by the store)
  float s_dq;        // the scale the de-quantised output uses
  float so, rs;      // SIM: scale (opaque copy) and RN(1 / scale); kernel: rs = 1 / scale (0 for an all-zero group)
  float qlo, qmax;
};

template <bool SIM>
__device__ __forceinline__ GroupScale group_scale(float amax, bool keeper, float clip) {
  GroupScale g;
  g.qmax = keeper ? 127.f : 7.f;
  g.qlo = keeper ? -128.f : -8.f;
  const float c = keeper ? 1.0f : clip;
  if constexpr (SIM) {
    amax = fmaxf(amax, (float)(half_t)1e-5f);               // quant.py:141-142
    if (c < 1.0f) amax = round_h(amax * c);                 // :168-169
    // amax / qmax and w / scales: correctly rounded FP32 quotients from  q1 = fma(fma(-q0,d,n), r, q0), q0 = n*r,
    // r = RN(1/d) -- exact for every finite fp16 n and positive fp16 d (tools/probes/div_probe.cpp; 7 and 127 are fp16)
    const float rq = keeper ? (1.0f / 127.0f) : (1.0f / 7.0f);
    const float a0 = opaque(amax);
    const float d0 = a0 * rq;
    const float s = round_h(__builtin_fmaf(__builtin_fmaf(-d0, g.qmax, a0), rq, d0));   // :170
    g.so = opaque(s);
    // RN(1/s) from v_rcp_f32 + one Newton step: equal to the IEEE quotient for every positive fp16 s (round_probe.cpp)
    const float r0 = __builtin_amdgcn_rcpf(g.so);
    g.rs = __builtin_fmaf(__builtin_fmaf(-r0, g.so, 1.0f), r0, r0);
    g.s_store = s;
    g.s_dq = s;
  } else {
    // Reorder.cuh:137-178
    if (c < 1.0f) amax = amax * c;
    const float sf = amax / g.qmax;
    g.so = sf;
    g.rs = sf != 0.f ? 1.0f / sf : 0.f;                     // all-zero group: codes 0 (0*inf = NaN in the reference)
    g.s_store = sf;
    g.s_dq = round_h(sf);
  }
  return g;
}

// the code of one value, as an integer-valued float
template <bool SIM>
__device__ __forceinline__ float group_code(float v, const GroupScale &g) {
  if constexpr (SIM) {
    const float q0 = v * g.rs;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, g.so, v), g.rs, q0);
    // :181 clamp(round(w / scales)): clamp first (bounds are integers), then round half-to-even
    return rintf(__builtin_amdgcn_fmed3f(round_h(q1), g.qlo, g.qmax));
  } else {
    const float t = __builtin_amdgcn_fmed3f(v * g.rs, g.qlo, g.qmax);
    // round half away from zero (CUDA round()) = sign(t) * floor(|t| + 0.5): v_cvt_rpi_i32_f32 computes
    // floor(x + 0.5) exactly (checked for every fp32 in [0, 300), tools/probes/round_probe.cpp)
    int ri;
    asm("v_cvt_rpi_i32_f32 %0, |%1|" : "=v"(ri) : "v"(t));
    return __builtin_copysignf((float)ri, t);
  }
}

// 16 codes (integer-valued floats) of one slot -> packed words, exact integer arithmetic in FP32: INT4 two's-complement nibbles,
// channel k of the slot at bits 4 (k % 8) of word k / 8 (2 words); INT8 keeper bytes, channel k at byte k % 4 of word k / 4 (4 words)
__device__ __forceinline__ v4u pack_codes16(const float (&tr)[16], bool keeper) {
  if (keeper) {
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float lo = __builtin_fmaf(tr[4 * k + 1], 256.f, tr[4 * k] + 32896.f);        // (c0+128) + (c1+128)*256
      const float hi = __builtin_fmaf(tr[4 * k + 3], 256.f, tr[4 * k + 2] + 32896.f);
      w[k] = ((unsigned)lo | ((unsigned)hi << 16)) ^ 0x80808080u;
    }
    return v4u{w[0], w[1], w[2], w[3]};
  }
  unsigned w[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float lo = 34952.f, hi = 34952.f;                                                    // sum 8*16^i, i<4
    lo = __builtin_fmaf(tr[8 * k + 0], 1.f, lo);
    lo = __builtin_fmaf(tr[8 * k + 1], 16.f, lo);
    lo = __builtin_fmaf(tr[8 * k + 2], 256.f, lo);
    lo = __builtin_fmaf(tr[8 * k + 3], 4096.f, lo);
    hi = __builtin_fmaf(tr[8 * k + 4], 1.f, hi);
    hi = __builtin_fmaf(tr[8 * k + 5], 16.f, hi);
    hi = __builtin_fmaf(tr[8 * k + 6], 256.f, hi);
    hi = __builtin_fmaf(tr[8 * k + 7], 4096.f, hi);
    w[k] = ((unsigned)lo | ((unsigned)hi << 16)) ^ 0x88888888u;
  }
  return v4u{w[0], w[1], 0u, 0u};
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float max8(float a) {   // max over the aligned 8 lanes this lane belongs to
  a = fmaxf(a, dpp_f<0xB1>(a));                      // quad_perm [1,0,3,2]
  a = fmaxf(a, dpp_f<0x4E>(a));                      // quad_perm [2,3,0,1]
  a = fmaxf(a, dpp_f<0x141>(a));                     // row_half_mirror
  return a;
}

// silu(a) * b for fp16 inputs given as floats.  Activate.cuh:28  x / (1 + expf(-x)) with the hardware exp2 / rcp (1 ulp each):
// 5 instructions instead of ~20; expf differs by ulps between libraries anyway (the parity tests allow codes +-1 on < 0.2 %)
template <bool SIM>
__device__ __forceinline__ float silu_mul(float a, float b) {
  const float e = __builtin_amdgcn_exp2f(a * -1.4426950408889634f);
  const float s = a * __builtin_amdgcn_rcpf(1.0f + e);
  if constexpr (SIM) return round_h(round_h(s) * b);        // act_fn(gate) * up, both in half
  else return s * b;                                        // kept in FP32 (Activate.cuh:103-106)
}

}  // namespace atom
