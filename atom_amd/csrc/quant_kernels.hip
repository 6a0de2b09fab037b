// Fused dynamic per-token activation quantisation for Atom on gfx950 (wave64).
//
//   atom_reorder_quant_f16          <- run_reorder_fp16_i4   (reference kernels/include/Reorder/Reorder.cuh:64-228)
//   atom_rmsnorm_reorder_quant_f16  <- run_rmsnorm_fp16_i4   (kernels/include/RMSNorm/RMSNorm.cuh:66-285)
//   atom_silu_mul_quant_f16         <- run_activate_fp16_i4  (kernels/include/Activate/Activate.cuh:67-217)
//
// All three are HBM-bound (2*H bytes read, ~H/2 written per token).  Design for CDNA4:
//   * one 256-thread workgroup (4 waves) per token row; the row is staged once into LDS with 16-byte
//     coalesced loads, the channel gather (reorder_index) then runs out of LDS, never out of HBM;
//   * 16 lanes own one 128-channel quantisation group (8 contiguous reordered channels per lane), so
//     the absmax reduction is 4 xor-shuffles inside a 16-lane row and every store is a 4/8/16-byte
//     word of a fully contiguous 64/128/256-byte run per group;
//   * runtime hidden size (any multiple of 128 up to 16384) instead of the reference's compile-time
//     4096 / 11008.
// Arithmetic is specified to the bit (see oracle/atom_oracle.py): ATOM_QUANT_SIM follows
// model/quant.py:141-181 (FP16 opmath), ATOM_QUANT_KERNEL follows Reorder.cuh:137-178 (FP32).
#include "common.h"

namespace atom {

enum ActOp { OP_REORDER = 0, OP_RMSNORM = 1, OP_SILU_MUL = 2 };

struct ActQuantParams {
  const half_t *x;        // reorder / rmsnorm: [M,H];  silu_mul: a [M,H]
  const half_t *b;        // rmsnorm: weight [H];       silu_mul: b [M,H]
  const int16_t *idx;     // reorder index [H] (reorder / rmsnorm)
  int64_t M;
  int H;
  int sim;                // 1 = ATOM_QUANT_SIM
  float clip;
  float eps;
  int ref_layout;         // 1 = ATOM_SCALE_LAYOUT_REF
  int64_t ld;             // halves between consecutive groups in norm_scales
  int8_t *o8;
  uint8_t *o4;
  half_t *s8;
  half_t *s4;
  half_t *xq;             // optional
};

__device__ __forceinline__ float round_half_away(float t) {
  float tr = truncf(t);
  if (fabsf(t - tr) >= 0.5f) tr += copysignf(1.0f, t);
  return tr;
}

// Quantise the 8 values this lane owns; the 16 lanes of a group cooperate on the absmax.
// Returns codes in q[], the scale to store (as float, exact value of the stored half in SIM mode) and
// the de-quantised values (float, to be rounded to half by the caller).
template <bool SIM, bool DQ>
__device__ __forceinline__ void quant_tail(const float (&v)[8], bool keeper, float clip, float &scale_store,
                                           int (&q)[8], float (&dq)[8]) {
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
  const float qmax = keeper ? 127.f : 7.f;
  const int qhi = keeper ? 127 : 7, qlo = keeper ? -128 : -8;
  const float c = keeper ? 1.0f : clip;
  if constexpr (SIM) {
    // quant.py:141-142: w.abs().amax().clamp(min=1e-5)  (the scalar is cast to half)
    amax = fmaxf(amax, (float)(half_t)1e-5f);
    if (c < 1.0f) amax = round_h(amax * c);                 // :168-169
    const float s = round_h(opaque(amax) / qmax);           // :170
    const float so = opaque(s);
    // w / scales with ONE IEEE divide per group: q1 = fma(fma(-q0,s,v), r, q0), q0 = v*r, r = 1/s is the correctly
    // rounded FP32 quotient for EVERY finite fp16 v and positive fp16 s (all 2.0e9 pairs checked on gfx950,
    // tools/probes/div_probe.cpp; only -0/s comes out as +0, which quantises to the same code)
    const float rs = 1.0f / so;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float q0 = v[i] * rs;
      const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, so, v[i]), rs, q0);
      const float t = rintf(round_h(q1));                   // :181 torch.round(w / scales), half
      // clamp in the integer domain (one v_med3_i32); |t| <= 65504/1e-6 fits int32 only after saturation, so
      // saturate in float first when it could overflow (never for sane data; keeps the cast defined)
      q[i] = min(max((int)fminf(fmaxf(t, -1e9f), 1e9f), qlo), qhi);
      if constexpr (DQ) dq[i] = (float)q[i] * s;           // (q + 0) * s: a code of -0.0 de-quantises to +0.0
    }
    scale_store = s;
  } else {
    // Reorder.cuh:137-178
    if (c < 1.0f) amax = amax * c;
    const float sf = amax / qmax;
    const float r = 1.0f / sf;
    const float sh = round_h(sf);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float t = round_half_away(v[i] * r);
      if (!(sf != 0.f)) t = 0.f;                            // all-zero group: 0*inf = NaN in the reference; codes 0 here
      q[i] = min(max((int)fminf(fmaxf(t, -1e9f), 1e9f), qlo), qhi);
      if constexpr (DQ) dq[i] = (float)q[i] * sh;
    }
    scale_store = sf;
  }
}

template <int OP, bool SIM, bool DQ>
__global__ __launch_bounds__(256) void act_quant_kernel(ActQuantParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t *row = reinterpret_cast<half_t *>(smem);           // H halves (unused for SILU_MUL)

  const int tid = threadIdx.x;
  const int64_t r = blockIdx.x;
  const int H = p.H;
  const half_t *xrow = p.x + r * (int64_t)H;

  if constexpr (OP != OP_SILU_MUL) {
    double ss = 0.0;
    for (int i = tid * 8; i < H; i += 256 * 8) {
      v4u raw = *reinterpret_cast<const v4u *>(xrow + i);
      *reinterpret_cast<v4u *>(row + i) = raw;
      if constexpr (OP == OP_RMSNORM) {
        const half_t *hv = reinterpret_cast<const half_t *>(&raw);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const double d = (double)hv[k];
          ss += d * d;
        }
      }
    }
    if constexpr (OP == OP_RMSNORM) {
      // sum of squares in FP64 (fp16 squares are exact), rounded to FP32 once -> order-independent
      double *red = reinterpret_cast<double *>(smem + ((H * 2 + 15) & ~15));
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
      if ((tid & 63) == 0) red[tid >> 6] = ss;
      __syncthreads();
      const double tot = ((red[0] + red[1]) + red[2]) + red[3];
      const float var = (float)(tot / (double)H);
      const float rinv = 1.0f / sqrtf(var + p.eps);         // correctly rounded sqrt and divide
      for (int i = tid * 8; i < H; i += 256 * 8) {
        v4u raw = *reinterpret_cast<v4u *>(row + i);        // this thread's own stores
        v4u wraw = *reinterpret_cast<const v4u *>(p.b + i);
        half_t *hv = reinterpret_cast<half_t *>(&raw);
        const half_t *wv = reinterpret_cast<const half_t *>(&wraw);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float xf = (float)hv[k], wf = (float)wv[k];
          if constexpr (SIM) {
            // HF LlamaRMSNorm: half(x * rsqrt(var+eps)) then weight * that, in half
            hv[k] = f2h(wf * round_h(xf * rinv));
          } else {
            // RMSNorm.cuh:145-151: half(float(x) * float(w) * r)
            hv[k] = f2h((xf * wf) * rinv);
          }
        }
        *reinterpret_cast<v4u *>(row + i) = raw;
      }
    }
    __syncthreads();
  }

  const int Gt = H >> 7;                 // groups incl. the keeper (last)
  const int K4h = (H - kKeeper) >> 1;    // packed bytes per row
  const int j = tid & 15;                // lane's slot inside the group
  for (int g = tid >> 4; g < Gt; g += 16) {
    const bool keeper = (g == Gt - 1);
    const int e0 = g * kGroup + j * 8;
    float v[8];
    if constexpr (OP == OP_SILU_MUL) {
      v4u ra = *reinterpret_cast<const v4u *>(xrow + e0);
      v4u rb = *reinterpret_cast<const v4u *>(p.b + r * (int64_t)H + e0);
      const half_t *av = reinterpret_cast<const half_t *>(&ra);
      const half_t *bv = reinterpret_cast<const half_t *>(&rb);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float a = (float)av[k];
        float s = a / (1.0f + expf(-a));                    // Activate.cuh:28
        if constexpr (SIM) {
          v[k] = round_h(round_h(s) * (float)bv[k]);        // act_fn(gate) * up, both in half
        } else {
          v[k] = s * (float)bv[k];                          // kept in FP32 (Activate.cuh:103-106)
        }
      }
    } else {
      if (p.idx) {
        v4u ri = *reinterpret_cast<const v4u *>(p.idx + e0);
        const uint16_t *iv = reinterpret_cast<const uint16_t *>(&ri);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (float)row[iv[k]];
      } else {                                   // identity order (input already reordered)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (float)row[e0 + k];
      }
    }

    int q[8];
    float dq[8];
    float sc;
    quant_tail<SIM, DQ>(v, keeper, p.clip, sc, q, dq);

    if (keeper) {
      v2u w;
      w.x = (q[0] & 0xFF) | ((q[1] & 0xFF) << 8) | ((q[2] & 0xFF) << 16) | ((unsigned)(q[3] & 0xFF) << 24);
      w.y = (q[4] & 0xFF) | ((q[5] & 0xFF) << 8) | ((q[6] & 0xFF) << 16) | ((unsigned)(q[7] & 0xFF) << 24);
      *reinterpret_cast<v2u *>(p.o8 + r * kKeeper + j * 8) = w;
    } else {
      unsigned w = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) w |= (unsigned)(q[k] & 0xF) << (4 * k);
      *reinterpret_cast<unsigned *>(p.o4 + r * (int64_t)K4h + g * 64 + j * 4) = w;
    }
    if (j == 0) {
      half_t *dst = keeper ? p.s8 : (p.s4 + (int64_t)g * p.ld);
      const half_t sh = f2h(sc);
      if (p.ref_layout) {
        const int base = ref_scale_index((int)r);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[base + 2 * k] = sh;
      } else {
        dst[r] = sh;
      }
    }
    if constexpr (DQ) {
      v4u o;
      half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
      for (int k = 0; k < 8; ++k) ov[k] = f2h(dq[k]);
      *reinterpret_cast<v4u *>(p.xq + r * (int64_t)H + e0) = o;
    }
  }
}

static int launch_act_quant(int op, ActQuantParams p, int quant_mode, int scale_layout, void *stream) {
  if (!p.x || !p.o8 || !p.o4 || !p.s8 || !p.s4) return ATOM_ERR_INVALID_ARG;
  if (op != OP_REORDER && !p.b) return ATOM_ERR_INVALID_ARG;
  if (quant_mode != ATOM_QUANT_KERNEL && quant_mode != ATOM_QUANT_SIM) return ATOM_ERR_INVALID_ARG;
  if (scale_layout != ATOM_SCALE_LAYOUT_REF && scale_layout != ATOM_SCALE_LAYOUT_PLAIN)
    return ATOM_ERR_INVALID_ARG;
  if (p.M < 1 || p.M > 0x7fffffff || p.H < 256 || p.H > 16384 || (p.H % kGroup) != 0) return ATOM_ERR_SHAPE;
  if (!aligned16(p.x) || !aligned16(p.b) || !aligned16(p.idx) || !aligned16(p.o8) || !aligned16(p.o4) ||
      !aligned16(p.xq))
    return ATOM_ERR_ALIGN;
  p.sim = quant_mode == ATOM_QUANT_SIM;
  p.ref_layout = scale_layout == ATOM_SCALE_LAYOUT_REF;
  p.ld = (int64_t)atom_scale_size(p.M, scale_layout);
  const size_t lds = op == OP_SILU_MUL ? 0 : (size_t)((p.H * 2 + 15) & ~15) + 64;
  dim3 grid((unsigned)p.M), block(256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define ATOM_LAUNCH(OPV)                                                                   \
  if (p.sim && p.xq)                                                                       \
    hipLaunchKernelGGL((act_quant_kernel<OPV, true, true>), grid, block, lds, s, p);       \
  else if (p.sim)                                                                          \
    hipLaunchKernelGGL((act_quant_kernel<OPV, true, false>), grid, block, lds, s, p);      \
  else if (p.xq)                                                                           \
    hipLaunchKernelGGL((act_quant_kernel<OPV, false, true>), grid, block, lds, s, p);      \
  else                                                                                     \
    hipLaunchKernelGGL((act_quant_kernel<OPV, false, false>), grid, block, lds, s, p);
  switch (op) {
    case OP_REORDER: ATOM_LAUNCH(OP_REORDER) break;
    case OP_RMSNORM: ATOM_LAUNCH(OP_RMSNORM) break;
    default: ATOM_LAUNCH(OP_SILU_MUL) break;
  }
#undef ATOM_LAUNCH
  return check_launch();
}

// ------------------------------------------------------------------------------------------------
// Weight quantise + pack: one wave per (channel_group rows x 128 columns) block.
//   reference: QLinearLayer.quant (model/qLinearLayer.py:42-78) + quantize_tensor_channel_group
//   (model/quant.py:68-107).  Offline path; not tuned.
struct WeightQuantParams {
  const half_t *W;
  int64_t N;
  int K;
  float clip;
  int cg;
  uint8_t *B4;
  int8_t *B8;
  half_t *sB;
  half_t *sB8;
  half_t *Wq;
};

__global__ __launch_bounds__(256) void weight_quant_kernel(WeightQuantParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t pair = blockIdx.x;                 // rows 2*pair, 2*pair+1
  const int K = p.K;
  const int K4 = K - kKeeper;
  const int G = K4 >> 7;
  const int64_t n = pair * 2 + (lane >> 5);
  const int c0 = (lane & 31) * 4;
  for (int g = wave; g <= G; g += 4) {
    const bool keeper = (g == G);
    const half_t *src = p.W + n * (int64_t)K + g * kGroup + c0;
    v2u raw = *reinterpret_cast<const v2u *>(src);
    const half_t *hv = reinterpret_cast<const half_t *>(&raw);
    float v[4];
    float amax = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = (float)hv[k];
      amax = fmaxf(amax, fabsf(v[k]));
    }
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
    if (!keeper && p.cg == 2) amax = fmaxf(amax, __shfl_xor(amax, 32));   // two rows share a scale
    const float qmax = keeper ? 127.f : 7.f, qmin = keeper ? -128.f : -8.f;
    const float c = keeper ? 1.0f : p.clip;
    amax = fmaxf(amax, (float)(half_t)1e-5f);
    if (c < 1.0f) amax = round_h(amax * c);
    const float s = round_h(opaque(amax) / qmax);
    const float so = opaque(s);
    const float rs = 1.0f / so;
    int q[4];
    v2u o;
    half_t *ov = reinterpret_cast<half_t *>(&o);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float q0 = v[k] * rs;
      float t = rintf(round_h(__builtin_fmaf(__builtin_fmaf(-q0, so, v[k]), rs, q0)));
      t = fminf(fmaxf(t, qmin), qmax);
      q[k] = (int)t;
      ov[k] = f2h((float)q[k] * s);
    }
    if (keeper) {
      const unsigned w = (q[0] & 0xFF) | ((q[1] & 0xFF) << 8) | ((q[2] & 0xFF) << 16) | ((unsigned)(q[3] & 0xFF) << 24);
      *reinterpret_cast<unsigned *>(p.B8 + n * kKeeper + c0) = w;
      if ((lane & 31) == 0) p.sB8[n] = f2h(s);
    } else {
      const unsigned short w = (unsigned short)((q[0] & 0xF) | ((q[1] & 0xF) << 4) | ((q[2] & 0xF) << 8) | ((q[3] & 0xF) << 12));
      *reinterpret_cast<unsigned short *>(p.B4 + n * (int64_t)(K4 >> 1) + g * 64 + (c0 >> 1)) = w;
      if ((lane & 31) == 0) p.sB[(int64_t)g * p.N + n] = f2h(s);
    }
    if (p.Wq) *reinterpret_cast<v2u *>(p.Wq + n * (int64_t)K + g * kGroup + c0) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// Pack an ALREADY fake-quantised weight: recover (codes, fp16 scale) per (channel_group rows x 128 columns) block of
// a weight whose values are half(s*c), c in [-8,7] (INT4 part) / [-128,127] (last 128 columns, one scale per row).
// This is what GPTQ leaves behind: gptq.py:38-39 computes q = scale*(clamp(round(x/scale)+zero,0,maxq)-zero) with an
// FP32 scale found on the error-compensated weight (gptq.py:285-287) and stores half(q) (gptq.py:331); the scale
// itself is thrown away.  Same wave geometry as weight_quant_kernel.  Per block:
//   for k = kmax..1 (the code magnitude of the largest |v|):  s = amax/k, c_i = rint(v_i/s); accept k when every c_i
//   is in range and |c_i*s - v_i| <= tol(v_i) = 2^-9|v_i| + 2^-24 (half(s32*c) carries 2^-11 relative rounding, so
//   does amax); refine s by least squares (sum v c / sum c^2), try the 5 fp16 values around it and keep the
//   (k, s) whose half(c*s) reproduces v best -- stopping at the first EXACT reproduction (always exists for weights
//   written by QLinearLayer.quant, whose scale is fp16).  A block with no acceptable k is re-quantised round-to-nearest
//   and counted in *bad.  Offline path; not tuned.
struct WeightPackParams {
  const half_t *W;
  int64_t N;
  int K;
  int cg;
  uint8_t *B4;
  int8_t *B8;
  half_t *sB;
  half_t *sB8;
  int *bad;
};

__global__ __launch_bounds__(256) void weight_pack_kernel(WeightPackParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t pair = blockIdx.x;
  const int K = p.K;
  const int K4 = K - kKeeper;
  const int G = K4 >> 7;
  const int64_t n = pair * 2 + (lane >> 5);
  const int c0 = (lane & 31) * 4;
  for (int g = wave; g <= G; g += 4) {
    const bool keeper = (g == G);
    const bool whole = !keeper && p.cg == 2;              // the unit is the whole wave (two rows), else one 32-lane half
    const half_t *src = p.W + n * (int64_t)K + g * kGroup + c0;
    v2u raw = *reinterpret_cast<const v2u *>(src);
    const half_t *hv = reinterpret_cast<const half_t *>(&raw);
    float v[4], tol[4];
    float amax = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = (float)hv[k];
      amax = fmaxf(amax, fabsf(v[k]));
      tol[k] = fabsf(v[k]) * 0x1p-9f + 0x1p-24f;
    }
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
    if (whole) amax = fmaxf(amax, __shfl_xor(amax, 32));
    const unsigned long long umask = whole ? ~0ull : (lane < 32 ? 0xFFFFFFFFull : 0xFFFFFFFF00000000ull);
    const float qmax = keeper ? 127.f : 7.f, qmin = keeper ? -128.f : -8.f;
    const int kmax = keeper ? 128 : 8;

    int best_q[4] = {0, 0, 0, 0};
    float best_s = 0.f, best_err = 3.0e38f;               // err: max_i |half(c*s)-v| / tol_i ; 0 = exact
    bool done = (amax == 0.f);                            // all-zero block: codes 0, scale 0
    if (done) best_err = 0.f;
    for (int k = kmax; k >= 1; --k) {
      if (__ballot(!done) == 0ull) break;                 // both units finished
      const float s = amax / (float)k;
      int q[4];
      bool ok = !done;
      float num = 0.f, den = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float t = rintf(v[i] / s);
        ok = ok && t >= qmin && t <= qmax && fabsf(t * s - v[i]) <= tol[i];
        q[i] = (int)fminf(fmaxf(t, qmin), qmax);
        num += v[i] * (float)q[i];
        den += (float)(q[i] * q[i]);
      }
      const bool unit_ok = !done && ((__ballot(!ok) & umask) == 0ull);
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) {
        num += __shfl_xor(num, m);
        den += __shfl_xor(den, m);
      }
      if (whole) {
        num += __shfl_xor(num, 32);
        den += __shfl_xor(den, 32);
      }
      const half_t sh0 = (half_t)(den > 0.f ? num / den : s);
      const unsigned short b0 = __builtin_bit_cast(unsigned short, sh0);
#pragma unroll
      for (int d = 0; d < 5; ++d) {
        const int off = d == 0 ? 0 : (d & 1 ? -((d + 1) >> 1) : (d >> 1));     // 0,-1,+1,-2,+2 ulp
        const unsigned short bits = (unsigned short)((int)b0 + off);
        const float sc = (float)__builtin_bit_cast(half_t, bits);
        float e = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) e = fmaxf(e, fabsf(round_h((float)q[i] * sc) - v[i]) / tol[i]);
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) e = fmaxf(e, __shfl_xor(e, m));
        if (whole) e = fmaxf(e, __shfl_xor(e, 32));
        if (!(sc > 0.f) || !(sc < 65504.f)) e = 3.0e38f;
        if (unit_ok && e < best_err) {
          best_err = e;
          best_s = sc;
#pragma unroll
          for (int i = 0; i < 4; ++i) best_q[i] = q[i];
        }
      }
      if (unit_ok && best_err == 0.f) done = true;
    }
    if (best_err > 1.0f) {                                // not on any 16/256-level grid: plain round-to-nearest
      if ((lane & 31) == 0 && (whole ? lane == 0 : true) && p.bad) atomicAdd(p.bad, 1);
      best_s = round_h(fmaxf(amax, 1e-5f) / qmax);
#pragma unroll
      for (int i = 0; i < 4; ++i) best_q[i] = (int)fminf(fmaxf(rintf(v[i] / best_s), qmin), qmax);
    }
    const int *q = best_q;
    if (keeper) {
      const unsigned w = (q[0] & 0xFF) | ((q[1] & 0xFF) << 8) | ((q[2] & 0xFF) << 16) | ((unsigned)(q[3] & 0xFF) << 24);
      *reinterpret_cast<unsigned *>(p.B8 + n * kKeeper + c0) = w;
      if ((lane & 31) == 0) p.sB8[n] = (half_t)best_s;
    } else {
      const unsigned short w = (unsigned short)((q[0] & 0xF) | ((q[1] & 0xF) << 4) | ((q[2] & 0xF) << 8) | ((q[3] & 0xF) << 12));
      *reinterpret_cast<unsigned short *>(p.B4 + n * (int64_t)(K4 >> 1) + g * 64 + (c0 >> 1)) = w;
      if ((lane & 31) == 0) p.sB[(int64_t)g * p.N + n] = (half_t)best_s;
    }
  }
}

}  // namespace atom

using namespace atom;

extern "C" {

size_t atom_scale_size(int64_t rows, int scale_layout) {
  if (rows < 0) return 0;
  return scale_layout == ATOM_SCALE_LAYOUT_REF ? (size_t)ref_scale_size(rows) : (size_t)rows;
}

int atom_reorder_quant_f16(const void *x, const int16_t *reorder_index, int64_t M, int hidden, int quant_mode,
                           float clip, int scale_layout, void *o_outliers, void *o_norms, void *outlier_scales,
                           void *norm_scales, void *xq_f16, void *stream) {
  ActQuantParams p{};
  p.x = (const half_t *)x; p.idx = reorder_index; p.M = M; p.H = hidden; p.clip = clip;
  p.o8 = (int8_t *)o_outliers; p.o4 = (uint8_t *)o_norms; p.s8 = (half_t *)outlier_scales;
  p.s4 = (half_t *)norm_scales; p.xq = (half_t *)xq_f16;
  return launch_act_quant(OP_REORDER, p, quant_mode, scale_layout, stream);
}

int atom_rmsnorm_reorder_quant_f16(const void *x, const void *weight, float eps, const int16_t *reorder_index,
                                   int64_t M, int hidden, int quant_mode, float clip, int scale_layout,
                                   void *o_outliers, void *o_norms, void *outlier_scales, void *norm_scales,
                                   void *xq_f16, void *stream) {
  ActQuantParams p{};
  p.x = (const half_t *)x; p.b = (const half_t *)weight; p.eps = eps; p.idx = reorder_index; p.M = M;
  p.H = hidden; p.clip = clip;
  p.o8 = (int8_t *)o_outliers; p.o4 = (uint8_t *)o_norms; p.s8 = (half_t *)outlier_scales;
  p.s4 = (half_t *)norm_scales; p.xq = (half_t *)xq_f16;
  return launch_act_quant(OP_RMSNORM, p, quant_mode, scale_layout, stream);
}

int atom_silu_mul_quant_f16(const void *a, const void *b, int64_t M, int hidden, int quant_mode, float clip,
                            int scale_layout, void *o_outliers, void *o_norms, void *outlier_scales,
                            void *norm_scales, void *xq_f16, void *stream) {
  ActQuantParams p{};
  p.x = (const half_t *)a; p.b = (const half_t *)b; p.M = M; p.H = hidden; p.clip = clip;
  p.o8 = (int8_t *)o_outliers; p.o4 = (uint8_t *)o_norms; p.s8 = (half_t *)outlier_scales;
  p.s4 = (half_t *)norm_scales; p.xq = (half_t *)xq_f16;
  return launch_act_quant(OP_SILU_MUL, p, quant_mode, scale_layout, stream);
}

int atom_quant_weight_w4(const void *W_f16, int64_t N, int64_t K_total, float w_clip, int channel_group, void *B4,
                         void *B8, void *sB, void *sB8, void *Wq_f16, void *stream) {
  if (!W_f16 || !B4 || !B8 || !sB || !sB8) return ATOM_ERR_INVALID_ARG;
  if (channel_group != 1 && channel_group != 2) return ATOM_ERR_INVALID_ARG;
  if (N < 2 || (N % 2) != 0 || K_total < 256 || (K_total % kGroup) != 0 || K_total > (1 << 20)) return ATOM_ERR_SHAPE;
  if (!aligned16(W_f16) || !aligned16(B4) || !aligned16(B8) || !aligned16(Wq_f16)) return ATOM_ERR_ALIGN;
  WeightQuantParams p{(const half_t *)W_f16, N, (int)K_total, w_clip, channel_group, (uint8_t *)B4,
                      (int8_t *)B8,          (half_t *)sB, (half_t *)sB8, (half_t *)Wq_f16};
  hipLaunchKernelGGL(weight_quant_kernel, dim3((unsigned)(N / 2)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), p);
  return check_launch();
}

int atom_pack_weight_w4(const void *Wq_f16, int64_t N, int64_t K_total, int channel_group, void *B4, void *B8,
                        void *sB, void *sB8, int32_t *bad_blocks, void *stream) {
  if (!Wq_f16 || !B4 || !B8 || !sB || !sB8) return ATOM_ERR_INVALID_ARG;
  if (channel_group != 1 && channel_group != 2) return ATOM_ERR_INVALID_ARG;
  if (N < 2 || (N % 2) != 0 || K_total < 256 || (K_total % kGroup) != 0 || K_total > (1 << 20)) return ATOM_ERR_SHAPE;
  if (!aligned16(Wq_f16) || !aligned16(B4) || !aligned16(B8)) return ATOM_ERR_ALIGN;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (bad_blocks && hipMemsetAsync(bad_blocks, 0, sizeof(int32_t), s) != hipSuccess) return ATOM_ERR_LAUNCH;
  WeightPackParams p{(const half_t *)Wq_f16, N, (int)K_total, channel_group, (uint8_t *)B4,
                     (int8_t *)B8,           (half_t *)sB, (half_t *)sB8, (int *)bad_blocks};
  hipLaunchKernelGGL(weight_pack_kernel, dim3((unsigned)(N / 2)), dim3(256), 0, s, p);
  return check_launch();
}

}  // extern "C"
