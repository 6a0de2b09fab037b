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

// 1.0f / sqrtf(x), BOTH operations correctly rounded (what hipcc's default expansion delivers and the oracle restates), without the
// range handling of that expansion: for 2^-64 <= x <= 2^64 -- every RMSNorm statistic `mean(x^2) + eps` of fp16 data with a sane eps --
// v_sqrt_f32's result is off by at most one ulp and the two residual tests below pick the correctly rounded neighbour (the compiler's
// own fix-up minus its 2^32 pre-scaling for tiny arguments), and the reciprocal is the compiler's Newton / residual sequence with the
// v_div_scale / v_div_fmas / v_div_fixup wrappers dropped (they are identities for a numerator of 1 and a denominator in [2^-32, 2^32]).
// 17 instructions instead of 45 per wave and row of the RMSNorm quantisers; outside the range the generic expression runs.  Checked
// against the generic expression for 1.5e9 arguments on the hardware (tools/probes/rinv_probe.cpp, profiles/r05/rinv_probe.txt).
__device__ __forceinline__ float rinv_sqrt_exact(float x) {
  if (!(x >= 0x1p-64f && x <= 0x1p+64f)) return 1.0f / sqrtf(x);   // (wave-uniform in the kernels: x is a row statistic)
  float y = __builtin_amdgcn_sqrtf(x);
  const float ym = __int_as_float(__float_as_int(y) - 1), yp = __int_as_float(__float_as_int(y) + 1);
  const float em = __builtin_fmaf(-ym, y, x), ep = __builtin_fmaf(-yp, y, x);
  y = em <= 0.f ? ym : y;
  y = ep > 0.f ? yp : y;
  float r = __builtin_amdgcn_rcpf(y);
  r = __builtin_fmaf(__builtin_fmaf(-y, r, 1.0f), r, r);
  float q = r;                                              // 1.0f * r
  q = __builtin_fmaf(__builtin_fmaf(-y, q, 1.0f), r, q);
  return __builtin_fmaf(__builtin_fmaf(-y, q, 1.0f), r, q);
}

// ------------------------------------------------------------------------------------------------------------------------------
// SIM mode in the FP16 domain (round 4).  The simulated path's values ARE halves (model/quant.py:134-181 runs in fp16), yet round 1-3
// carried them as floats: per value 3 FP32 ops for the exact quotient + 2 conversions to round it to half + clamp + rint = 7, after
// 3 for the RMSNorm product -- 112 + 48 of the 360 VALU a wave spends on a row (profiles/r04/pmc_quantisers.txt: the kernels are
// VALU-bound).  Here a slot's 16 values are 8 HALF PAIRS and the work that can be packed is:
//   pair i of a slot = channels (L, L + 4), L = 8 (i / 4) + i % 4   -- the nibble order of the packed word falls out of the pairing
//   q0 = half(v * RN(1/s))                      v_fma_mixlo/hi_f16   (any faithful estimate will do)
//   e  = v - s * q0          exactly, in FP32   v_fma_mix_f32        (22-bit product, cancellation: fits 24 bits)
//   q  = half(q0 + e * RN(1/s))                 v_fma_mixlo/hi_f16   == RN_half(v / s): q0 + e r = (v / s)(1 + 2^-25), and a quotient of
//        two halves that is not itself a half or a midpoint is >= 2^-23 (relative) away from the nearest midpoint of two halves
//   clamp, then + 1536                          v_pk_max/min/add_f16 (half(q + 1536) IS round-half-even to an integer; the code sits in
//        the low bits of the half's bit pattern, two's complement: 0x6600 + c)
// = 4.5 per value and the result is already packed pairwise.  Bit-identical to group_code<true> for every half v and every scale a
// group can have with clip >= 1/64 (smaller clips could overflow q0 in FP16: the callers take the FP32 form then); checked against
// the oracle by the same tests as before.  The sequences are inline asm: hipcc turns the C form into packed FP32 + conversions, and
// dependent 16-bit partial writes need a wait state it cannot see inside asm -- every block keeps dependent instructions >= 2 apart.
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__host__ __device__ constexpr int pair_lo(int i) { return 8 * (i >> 2) + (i & 3); }          // pair i = channels (pair_lo, pair_lo + 4)
constexpr float kSimHalfMinClip = 1.0f / 64.0f;

#define ATOM_Q0LO(i) "v_fma_mixlo_f16 %[q" #i "], %[z" #i "], %[rs], 0 op_sel_hi:[1,0,0]\n\t"
#define ATOM_Q0HI(i) "v_fma_mixhi_f16 %[q" #i "], %[z" #i "], %[rs], 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
#define ATOM_ELO(i) "v_fma_mix_f32 %[a" #i "], -%[s], %[q" #i "], %[z" #i "] op_sel_hi:[1,1,1]\n\t"
#define ATOM_EHI(i) "v_fma_mix_f32 %[b" #i "], -%[s], %[q" #i "], %[z" #i "] op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"
#define ATOM_Q1LO(i) "v_fma_mixlo_f16 %[t" #i "], %[a" #i "], %[rs], %[q" #i "] op_sel_hi:[0,0,1]\n\t"
#define ATOM_Q1HI(i) "v_fma_mixhi_f16 %[t" #i "], %[b" #i "], %[rs], %[q" #i "] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
#define ATOM_CMAX(i) "v_pk_max_f16 %[t" #i "], %[t" #i "], %[lo]\n\t"
#define ATOM_CMIN(i) "v_pk_min_f16 %[t" #i "], %[t" #i "], %[hi]\n\t"
#define ATOM_CADD(i) "v_pk_add_f16 %[t" #i "], %[t" #i "], %[mg]\n\t"
#define ATOM_X4(M) M(0) M(1) M(2) M(3)
// four pairs: t = half pairs (1536 + code).  s2 = the scale (half) in the low 16 bits; lo2 / hi2 = the clamp bounds in both halves
__device__ __forceinline__ void sim_codes4(unsigned z0, unsigned z1, unsigned z2, unsigned z3, float rs, unsigned s2, unsigned lo2,
                                           unsigned hi2, unsigned &t0, unsigned &t1, unsigned &t2, unsigned &t3) {
  unsigned q0, q1, q2, q3;
  float a0, a1, a2, a3, b0, b1, b2, b3;
  const unsigned mg2 = 0x66006600u;                                    // (1536, 1536)
  asm(ATOM_X4(ATOM_Q0LO) ATOM_X4(ATOM_Q0HI) ATOM_X4(ATOM_ELO) ATOM_X4(ATOM_EHI) ATOM_X4(ATOM_Q1LO) ATOM_X4(ATOM_Q1HI)
      ATOM_X4(ATOM_CMAX) ATOM_X4(ATOM_CMIN) ATOM_X4(ATOM_CADD)
      : [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3),
        [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
      : [z0] "v"(z0), [z1] "v"(z1), [z2] "v"(z2), [z3] "v"(z3), [rs] "v"(rs), [s] "v"(s2), [lo] "v"(lo2), [hi] "v"(hi2), [mg] "v"(mg2));
}
#undef ATOM_Q0LO
#undef ATOM_Q0HI
#undef ATOM_ELO
#undef ATOM_EHI
#undef ATOM_Q1LO
#undef ATOM_Q1HI
#undef ATOM_CMAX
#undef ATOM_CMIN
#undef ATOM_CADD

// max |.| over the 16 halves of 8 pairs, as a float (two interleaved v_max3_f16 chains: |.| is a free source modifier there)
__device__ __forceinline__ float amax16_h(const unsigned (&z)[8]) {
  unsigned ma, mb;
  float r;
  asm("v_max3_f16 %[ma], |%[z0]|, |%[z0]|, |%[z1]| op_sel:[0,1,0,0]\n\t"
      "v_max3_f16 %[mb], |%[z4]|, |%[z4]|, |%[z5]| op_sel:[0,1,0,0]\n\t"
      "v_max3_f16 %[ma], %[ma], |%[z1]|, |%[z2]| op_sel:[0,1,0,0]\n\t"
      "v_max3_f16 %[mb], %[mb], |%[z5]|, |%[z6]| op_sel:[0,1,0,0]\n\t"
      "v_max3_f16 %[ma], %[ma], |%[z2]|, |%[z3]| op_sel:[0,1,0,0]\n\t"
      "v_max3_f16 %[mb], %[mb], |%[z6]|, |%[z7]| op_sel:[0,1,0,0]\n\t"
      "v_max3_f16 %[ma], %[ma], |%[z3]|, |%[z7]| op_sel:[0,1,1,0]\n\t"
      "s_nop 0\n\t"
      "v_max_f16 %[ma], %[ma], %[mb]\n\t"
      "s_nop 0\n\t"
      "v_cvt_f32_f16 %[r], %[ma]\n\t"
      : [ma] "=&v"(ma), [mb] "=&v"(mb), [r] "=v"(r)
      : [z0] "v"(z[0]), [z1] "v"(z[1]), [z2] "v"(z[2]), [z3] "v"(z[3]), [z4] "v"(z[4]), [z5] "v"(z[5]), [z6] "v"(z[6]), [z7] "v"(z[7]));
  return r;
}

// max over the aligned 8 lanes, one v_max_f32 with a DPP operand per stage (fmaxf() around a DPP move costs a move and a
// canonicalising max more per stage); a DPP read of a just-written register needs two wait states
__device__ __forceinline__ float max8_dpp(float a) {
  asm("s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      : "+v"(a));
  return a;
}

// y = half(float(x) * rinv) for four pairs: pair i takes the low (HI = 0) or high (HI = 1) halves of xl[i] and xh[i]
template <int HI0, int HI1, int HI2, int HI3>
__device__ __forceinline__ void sim_scale4(unsigned xl0, unsigned xl1, unsigned xl2, unsigned xl3, unsigned xh0, unsigned xh1,
                                           unsigned xh2, unsigned xh3, float rinv, unsigned &y0, unsigned &y1, unsigned &y2,
                                           unsigned &y3) {
  asm("v_fma_mixlo_f16 %[y0], %[a0], %[r], 0 op_sel:[%c[h0],0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %[y1], %[a1], %[r], 0 op_sel:[%c[h1],0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %[y2], %[a2], %[r], 0 op_sel:[%c[h2],0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %[y3], %[a3], %[r], 0 op_sel:[%c[h3],0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %[y0], %[b0], %[r], 0 op_sel:[%c[h0],0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %[y1], %[b1], %[r], 0 op_sel:[%c[h1],0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %[y2], %[b2], %[r], 0 op_sel:[%c[h2],0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %[y3], %[b3], %[r], 0 op_sel:[%c[h3],0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 0\n\t"
      : [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3)
      : [a0] "v"(xl0), [a1] "v"(xl1), [a2] "v"(xl2), [a3] "v"(xl3), [b0] "v"(xh0), [b1] "v"(xh1), [b2] "v"(xh2), [b3] "v"(xh3),
        [r] "v"(rinv), [h0] "n"(HI0), [h1] "n"(HI1), [h2] "n"(HI2), [h3] "n"(HI3));
}

// silu(a) * b for fp16 inputs given as floats.  Activate.cuh:28  x / (1 + expf(-x)) with the hardware exp2 / rcp (1 ulp each):
// 5 instructions instead of ~20; expf differs by ulps between libraries anyway (the parity tests allow codes +-1 on < 0.2 %)
__device__ __forceinline__ float silu_f32(float a) {
  const float e = __builtin_amdgcn_exp2f(a * -1.4426950408889634f);
  return a * __builtin_amdgcn_rcpf(1.0f + e);
}
template <bool SIM>
__device__ __forceinline__ float silu_mul(float a, float b) {
  const float s = silu_f32(a);
  if constexpr (SIM) return round_h(round_h(s) * b);        // act_fn(gate) * up, both in half
  else return s * b;                                        // kept in FP32 (Activate.cuh:103-106)
}

}  // namespace atom
