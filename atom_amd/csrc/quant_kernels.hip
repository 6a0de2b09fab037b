// Fused dynamic per-token activation quantisation for Atom on gfx950 (wave64).
//
//   atom_reorder_quant_f16          <- run_reorder_fp16_i4   (reference kernels/include/Reorder/Reorder.cuh:64-228)
//   atom_rmsnorm_reorder_quant_f16  <- run_rmsnorm_fp16_i4   (kernels/include/RMSNorm/RMSNorm.cuh:66-285)
//   atom_silu_mul_quant_f16         <- run_activate_fp16_i4  (kernels/include/Activate/Activate.cuh:67-217)
//
// All three move 2*H (4*H for silu_mul) bytes in and ~H/2 out per token; on gfx950 the limiter is the VALU instruction
// count per element, not HBM (a first version with 8 elements per lane, ds_bpermute reductions and integer packing ran
// ~21 VALU per element and 2.1-3.3 TB/s; profiles/r01_actquant_ab.txt).  Design:
//   * 16 elements per lane, 8 lanes per 128-channel quantisation group: the absmax reduction is 3 DPP-modified v_max
//     (quad_perm, quad_perm, row_half_mirror), the per-group scale arithmetic is amortised over 16 elements, every
//     store is an 8/16/32-byte word of a fully contiguous 64/128/256-byte run per group;
//   * reorder / rmsnorm: persistent workgroups (256 threads).  The LDS byte offsets of a lane's channels (and its
//     gathered RMSNorm weights) are loop-invariant and live in registers; rows are staged by LDS-DMA (global_load_lds,
//     16 B per lane, no VGPR round trip) into a double buffer -- the next row's DMA is in flight while the current row
//     is quantised; the channel gather runs out of LDS, never out of HBM; RMSNorm normalises AFTER the gather (one LDS
//     pass); silu_mul needs no LDS at all (one row per workgroup, 32 B per lane and operand);
//   * codes: clamp (v_med3_f32) -> round -> one FMA per element into a base-16 / base-256 accumulator (bias folded
//     into the initial value, two's complement restored with one XOR per word); round-half-away is
//     sign * v_cvt_rpi_i32_f32(|t|); the exact FP16-opmath quotient is 3 FMAs around RN(1/s) = v_rcp_f32 + one Newton
//     step (both verified exhaustively on the hardware, tools/probes/round_probe.cpp, div_probe.cpp);
//   * de-quantised output: code*scale is exact in FP32, so the conversion to half is the single-rounding v_fma_mix;
//   * runtime hidden size (any multiple of 128 up to 16384) instead of the reference's compile-time 4096 / 11008.
// Arithmetic is specified to the bit (see oracle/atom_oracle.py): ATOM_QUANT_SIM follows
// model/quant.py:141-181 (FP16 opmath), ATOM_QUANT_KERNEL follows Reorder.cuh:137-178 (FP32).
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "quant_math.h"

namespace atom {

enum ActOp { OP_REORDER = 0, OP_RMSNORM = 1, OP_SILU_MUL = 2, OP_ADD_RMSNORM = 3 };

struct ActQuantParams {
  const half_t *x;        // reorder / rmsnorm: [M,H];  silu_mul: a [M,H]
  const half_t *b;        // rmsnorm: weight [H];       silu_mul: b [M,H]
  const half_t *res;      // add_rmsnorm: residual [M,H]
  half_t *res_out;        // add_rmsnorm: x + residual [M,H] (may alias res)
  const int16_t *idx;     // reorder index [H] (reorder / rmsnorm)
  int64_t M;
  int H;
  int sim;                // 1 = ATOM_QUANT_SIM
  int wide;               // 1 = ATOM_QUANT_WIDE_CODES: o4 is int8 [M, H-128] (code*16, even/odd de-interleaved per 32)
  int64_t f6_rows;        // > 0 = ATOM_QUANT_F6_CODES: o4 is uint8 [G][f6_rows][104] (BF6 stream + scale, atom_hip.h)
  float clip;
  float eps;
  int ref_layout;         // 1 = ATOM_SCALE_LAYOUT_REF
  int64_t ld;             // halves between consecutive groups in norm_scales
  int8_t *o8;
  uint8_t *o4;
  half_t *s8;
  half_t *s4;
  half_t *xq;             // optional
  int w_lds;              // rmsnorm kernels: 1 = the weight vector is staged in LDS (set by the launcher when it fits)
};

typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;

// What one slot (row r, group g, octet lane j: 16 reordered channels) writes: computed by slot_codes / slot_codes_h, written by
// slot_store -- apart, so that the persistent kernels can issue a row's stores AFTER the next iteration's wait (below).
template <bool DQ>
struct SlotOut {
  v4u w;                  // keeper: the 16 code bytes; packed / wide: the two nibble words in w[0], w[1]; BF6: the 12 stream bytes in w[0..2]
  unsigned sh;            // the group's scale as a half (low 16 bits)
  v4u o[DQ ? 2 : 1];      // DQ: the 16 de-quantised halves
};

// FMT: the code format as a compile-time constant (0 packed nibbles, 1 wide int8, 2 BF6 records) -- as run-time branches on
// p.wide / p.f6_rows the three store paths cost the packed path 4-7 % (register copies at the merges; round 4, same-box A/B)
template <bool DQ, int FMT>
__device__ __forceinline__ void slot_store(const SlotOut<DQ> &q, const ActQuantParams &p, int64_t r, int g, int j, int e0,
                                           bool keeper, int K4h) {
  typedef unsigned v3u __attribute__((ext_vector_type(3)));
  if (keeper) {
    *reinterpret_cast<v4u *>(p.o8 + r * kKeeper + j * 16) = q.w;
  } else {
    if constexpr (FMT == 2) {
      uint8_t *dst = p.o4 + ((int64_t)g * p.f6_rows + r) * 104;
      *reinterpret_cast<v3u *>(dst + 12 * j) = v3u{q.w[0], q.w[1], q.w[2]};
      if (j == 0) {                                       // the GEMM reads the token scale from the row itself: fp16 at byte 96,
        const float sf = (float)__builtin_bit_cast(half_t, (unsigned short)q.sh);   // the same value as fp32 at byte 100
        *reinterpret_cast<v2u *>(dst + 96) = v2u{q.sh, __builtin_bit_cast(unsigned, sf)};
      }
    } else if constexpr (FMT == 1) {
      // my 16 channels are half `j & 1` of 32-channel block g*4 + j/2: even channels -> chunk 0, odd -> chunk 1
      uint8_t *dst = p.o4 + r * (int64_t)(2 * K4h) + g * 128 + (j >> 1) * 32 + (j & 1) * 8;
      *reinterpret_cast<v2u *>(dst) = v2u{(q.w[0] << 4) & 0xF0F0F0F0u, (q.w[1] << 4) & 0xF0F0F0F0u};
      *reinterpret_cast<v2u *>(dst + 16) = v2u{q.w[0] & 0xF0F0F0F0u, q.w[1] & 0xF0F0F0F0u};
    } else {
      *reinterpret_cast<v2u *>(p.o4 + r * (int64_t)K4h + g * 64 + j * 8) = v2u{q.w[0], q.w[1]};
    }
  }
  if (j == 0) {
    half_t *dst = keeper ? p.s8 : (p.s4 + (int64_t)g * p.ld);
    const half_t sh = __builtin_bit_cast(half_t, (unsigned short)q.sh);
    if (p.ref_layout) {
      const int base = ref_scale_index((int)r);
#pragma unroll
      for (int k = 0; k < 4; ++k) dst[base + 2 * k] = sh;
    } else {
      dst[r] = sh;
    }
  }
  if constexpr (DQ) {
    v4u *dst = reinterpret_cast<v4u *>(p.xq + r * (int64_t)p.H + e0);
    dst[0] = q.o[0];
    dst[1] = q.o[1];
  }
}

// Quantise the 16 values of a slot (FP32 form: the kernel-flavoured mode, and the simulated mode's fallback)
template <bool SIM, bool DQ, int FMT>
__device__ __forceinline__ SlotOut<DQ> slot_codes(const float (&v)[16], const ActQuantParams &p, bool keeper) {
  SlotOut<DQ> q;
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[i]));
  amax = max8(amax);
  const GroupScale gs = group_scale<SIM>(amax, keeper, p.clip);
  float tr[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) tr[i] = group_code<SIM>(v[i], gs);
  q.sh = (unsigned)__builtin_bit_cast(unsigned short, f2h(gs.s_store));
  q.w = pack_codes16(tr, keeper);
  if constexpr (FMT == 2) {
    if (!keeper) {
      // BF6 (E3M2) holds every INT4 code exactly; v_cvt_scalef32_2xpk16_bf6_f32 converts AND packs 32 floats into 6-bit
      // fields, interleaving its two sources (field 2i = a[i], 2i+1 = b[i]; tools/probes): my 16 codes are fields 0..15
      typedef float v16f __attribute__((ext_vector_type(16)));
      typedef unsigned v6u __attribute__((ext_vector_type(6)));
      v16f ea, eb;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ea[i] = tr[2 * i];
        eb[i] = tr[2 * i + 1];
        ea[8 + i] = 0.f;
        eb[8 + i] = 0.f;
      }
      const v6u f = cvt_2xpk16_bf6(ea, eb);
      q.w = v4u{f[0], f[1], f[2], 0u};
    }
  }
  if constexpr (DQ) {
    // code * scale is exact in FP32 (8 x 11 significant bits), so one rounding to half == the reference's half
    // multiply; "+ 0" turns the -0 of a negative value that rounded to code 0 into the reference's +0
    half_t *ov = reinterpret_cast<half_t *>(q.o);
#pragma unroll
    for (int i = 0; i < 16; ++i) ov[i] = (half_t)__builtin_fmaf(tr[i], gs.s_dq, 0.0f);
  }
  return q;
}

template <bool SIM, bool DQ, int FMT>
__device__ __forceinline__ void quant_slot(const float (&v)[16], const ActQuantParams &p, int64_t r, int g, int j,
                                           int e0, bool keeper, int K4h) {
  slot_store<DQ, FMT>(slot_codes<SIM, DQ, FMT>(v, p, keeper), p, r, g, j, e0, keeper, K4h);
}

// The same for the simulated path with the slot as 8 half pairs (quant_math.h, "SIM mode in the FP16 domain"): z[i] = channels
// (pair_lo(i), pair_lo(i) + 4) of the slot.  Same codes, scales and de-quantised values as slot_codes<true, ...> bit for bit.
template <bool DQ, int FMT>
__device__ __forceinline__ SlotOut<DQ> slot_codes_h(const unsigned (&z)[8], const ActQuantParams &p, bool keeper) {
  SlotOut<DQ> q;
  const float amax = max8_dpp(amax16_h(z));
  const GroupScale gs = group_scale<true>(amax, keeper, p.clip);
  const unsigned s2 = (unsigned)__builtin_bit_cast(unsigned short, (half_t)gs.s_store);      // exact: s_store is a half value
  const unsigned lo2 = keeper ? 0xD800D800u : 0xC800C800u, hi2 = keeper ? 0x57F057F0u : 0x47004700u;   // (-128 | -8), (127 | 7)
  unsigned t[8];
  sim_codes4(z[0], z[1], z[2], z[3], gs.rs, s2, lo2, hi2, t[0], t[1], t[2], t[3]);
  sim_codes4(z[4], z[5], z[6], z[7], gs.rs, s2, lo2, hi2, t[4], t[5], t[6], t[7]);
  q.sh = s2;
  // t[i] = (0x6600 + c_lo, 0x6600 + c_hi): the codes are the low bits
  if (keeper) {
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const unsigned c01 = (t[4 * k] & 0x00FF00FFu) | ((t[4 * k + 1] & 0x00FF00FFu) << 8);      // low half: channels 8k + 0, 1; high: + 4, 5
      const unsigned c23 = (t[4 * k + 2] & 0x00FF00FFu) | ((t[4 * k + 3] & 0x00FF00FFu) << 8);
      w[2 * k] = __builtin_amdgcn_perm(c23, c01, 0x05040100u);                                  // channels 8k + 0..3
      w[2 * k + 1] = __builtin_amdgcn_perm(c23, c01, 0x07060302u);                              // channels 8k + 4..7
    }
    q.w = v4u{w[0], w[1], w[2], w[3]};
  } else if constexpr (FMT == 2) {
    typedef float v16f __attribute__((ext_vector_type(16)));
    typedef unsigned v6u __attribute__((ext_vector_type(6)));
    v16f ea, eb;                                          // field 2i = ea[i], 2i + 1 = eb[i]: channel 2i -> ea[i], 2i + 1 -> eb[i]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const h2v c = __builtin_bit_cast(h2v, t[i]) - h2v{(_Float16)1536.0f, (_Float16)1536.0f};
      const int L = pair_lo(i);                           // channels L and L + 4 (same parity)
      if (L & 1) { eb[L >> 1] = (float)c.x; eb[(L + 4) >> 1] = (float)c.y; }
      else { ea[L >> 1] = (float)c.x; ea[(L + 4) >> 1] = (float)c.y; }
    }
#pragma unroll
    for (int i = 8; i < 16; ++i) { ea[i] = 0.f; eb[i] = 0.f; }
    const v6u f = cvt_2xpk16_bf6(ea, eb);
    q.w = v4u{f[0], f[1], f[2], 0u};
  } else {
    unsigned w[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
      w[k] = (t[4 * k] & 0x000F000Fu) | ((t[4 * k + 1] & 0x000F000Fu) << 4) | ((t[4 * k + 2] & 0x000F000Fu) << 8) |
             ((t[4 * k + 3] & 0x000F000Fu) << 12);
    q.w = v4u{w[0], w[1], 0u, 0u};
  }
  if constexpr (DQ) {
    // code * scale rounded once to half (the reference's half multiply); codes come out of the subtraction as +0 for a negative
    // value that rounded to 0, as the FP32 form's "+ 0" arranges
    const h2v sv = {__builtin_bit_cast(_Float16, (unsigned short)s2), __builtin_bit_cast(_Float16, (unsigned short)s2)};
    unsigned d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      d[i] = __builtin_bit_cast(unsigned, (__builtin_bit_cast(h2v, t[i]) - h2v{(_Float16)1536.0f, (_Float16)1536.0f}) * sv);
#pragma unroll
    for (int k = 0; k < 2; ++k) {                         // dword m of the 16 halves = channels (2m, 2m + 1)
      q.o[k][0] = __builtin_amdgcn_perm(d[4 * k + 1], d[4 * k], 0x05040100u);       // channels 8k + 0, 1: low halves of pairs 0, 1
      q.o[k][1] = __builtin_amdgcn_perm(d[4 * k + 3], d[4 * k + 2], 0x05040100u);   // 8k + 2, 3
      q.o[k][2] = __builtin_amdgcn_perm(d[4 * k + 1], d[4 * k], 0x07060302u);       // 8k + 4, 5: high halves
      q.o[k][3] = __builtin_amdgcn_perm(d[4 * k + 3], d[4 * k + 2], 0x07060302u);   // 8k + 6, 7
    }
  }
  return q;
}

template <bool DQ, int FMT>
__device__ __forceinline__ void quant_slot_h(const unsigned (&z)[8], const ActQuantParams &p, int64_t r, int g, int j, int e0,
                                             bool keeper, int K4h) {
  slot_store<DQ, FMT>(slot_codes_h<DQ, FMT>(z, p, keeper), p, r, g, j, e0, keeper, K4h);
}

// reorder / rmsnorm: persistent workgroups, LDS-DMA double buffer.  NP = slots (16 channels) per thread per row.
// HC != 0: an instance for ONE hidden size (the launcher picks it when p.H == HC).  Everything derived from H folds, and -- what pays --
// the two row buffers sit at compile-time LDS offsets: the row loop is unrolled by two and a gathered channel's address is its
// loop-invariant offset register + an IMMEDIATE, where the generic form adds the buffer base to 16 offsets per thread and row
// (18 of the ~280 VALU instructions a wave spends on a row; round 5).
template <int OP, bool SIM, bool DQ, int NP, int FMT, int HC = 0>
__global__ __launch_bounds__(256) void act_quant2_kernel(ActQuantParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  {  // the kernel arguments in ONE batch of scalar loads (round 6: hipcc fetched them in two or three, a scalar-cache round trip apart;
     // at decode batches this kernel is a chain of such trips)
    const void *a0 = p.x, *a1 = p.b, *a2 = p.res, *a3 = p.res_out, *a4 = p.idx, *a5 = p.o8, *a6 = p.o4, *a7 = p.s8, *a8_ = p.s4, *a9 = p.xq;
    const int i0 = (int)p.M, i1 = p.H, i2 = p.ref_layout, i3 = p.w_lds, i4 = (int)gridDim.x, i5 = (int)p.f6_rows, i6 = (int)p.ld;
    asm volatile("" ::"s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(a6), "s"(a7), "s"(a8_), "s"(a9), "s"(i0), "s"(i1), "s"(i2), "s"(i3), "s"(i4), "s"(i5), "s"(i6));
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = HC ? HC : p.H;
  const int nslots = H >> 4;
  const int Gt = H >> 7;
  const int K4h = (H - kKeeper) >> 1;
  const int nchunks = H >> 3;                               // 16-byte chunks per row
  constexpr bool ADD = OP == OP_ADD_RMSNORM, NORM = OP == OP_RMSNORM || ADD;
  const int bufbytes = ((H * 2 + 1023) & ~1023);            // DMA blocks are 1 KiB (64 lanes x 16 B)
  const int stage = ADD ? 2 * bufbytes : bufbytes;          // ADD stages two rows: x, then the residual
  float *red = reinterpret_cast<float *>(smem + 2 * stage);        // [2][4] partial sums of squares
  const int j = tid & 7;

  const unsigned smem0 = lds_addr(smem);
  auto issue_row = [&](int64_t r, int b) {                  // DMA row r into buffer b
    const char *src = reinterpret_cast<const char *>(p.x + r * (int64_t)H);
    const char *src2 = reinterpret_cast<const char *>(p.res + r * (int64_t)H);
#pragma unroll
    for (int i = 0; i < 2 * NP; ++i) {
      const int blk = i * 4 + wave;                         // 1 KiB block
      if (blk * 64 < nchunks) {
        const int c = min(blk * 64 + lane, nchunks - 1);    // tail lanes re-read the last chunk into the padding
        lds_dma_sv<16>(src, (unsigned)c * 16u, smem0 + b * stage + blk * 1024);
        if constexpr (ADD)
          lds_dma_sv<16>(src2, (unsigned)c * 16u, smem0 + b * stage + bufbytes + blk * 1024);
      }
    }
  };

  // the first row is on its way before anything else: with one row per workgroup (decode batches) the kernel is a chain
  // of memory round trips, and the index / weight gathers below are two of them
  // Rows by XCD.  Neighbouring rows share cache lines in the outputs (F6 records 104 B apart, per-group scales 2 B apart) and every
  // XCD has its own L2, so rows interleaved over all workgroups leave eight L2s as partial lines.  Workgroup i runs on XCD i % 8
  // (round-robin dispatch): each XCD takes a contiguous eighth of the rows and its workgroups walk it side by side, so that
  // neighbours meet in ONE L2 within a row time (65,536 x 4096 in the F6 format: reorder 240 -> 215 us, RMSNorm 250 -> 218, and
  // SiLU x up below 313 -> 247; blocks of up to 16 consecutive rows per workgroup on top of this changed nothing).
  int64_t r = blockIdx.x, rstep = gridDim.x, rend = p.M;
  if (gridDim.x >= 64) {
    const int xcd = blockIdx.x & 7;
    const int64_t cm = (p.M + 7) >> 3;
    r = xcd * cm + (blockIdx.x >> 3);
    rstep = (gridDim.x - xcd + 7) >> 3;
    rend = min((int64_t)p.M, (xcd + 1) * cm);
  }
  int b = 0;
  if (r < rend) issue_row(r, 0);
  // RMSNorm weights: the whole vector rides into LDS with the first row (one more LDS-DMA stream); the per-channel gather below then
  // runs out of LDS instead of issuing 16 * NP scattered 2-byte global loads per thread behind the index loads (round 1: a second
  // dependent memory round trip before the first row could be touched -- most of the kernel at M <= 4096)
  char *wbuf = smem + 2 * stage + 1024;
  if constexpr (NORM) {
    if (p.w_lds) {
#pragma unroll
      for (int i = 0; i < 2 * NP; ++i) {
        const int blk = i * 4 + wave;
        if (blk * 64 < nchunks) {
          const int c = min(blk * 64 + lane, nchunks - 1);
          lds_dma_sv<16>(p.b, (unsigned)c * 16u, lds_addr(wbuf) + blk * 1024);
        }
      }
    }
  }

  // loop-invariant per-thread state: LDS byte offsets of my channels, gathered RMSNorm weights
  int off[NP][16];
  half_t wg[NP][16];                                        // (halves: 8 registers per slot, and the SIM product is a half multiply)
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int slot = ps * 256 + tid;
    const int e0 = min(slot, nslots - 1) * 16;
    if (p.idx) {
      v4u ri[2];
      ri[0] = *reinterpret_cast<const v4u *>(p.idx + e0);
      ri[1] = *reinterpret_cast<const v4u *>(p.idx + e0 + 8);
      const uint16_t *iv = reinterpret_cast<const uint16_t *>(ri);
#pragma unroll
      for (int k = 0; k < 16; ++k) off[ps][k] = (int)iv[k] * 2;
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) off[ps][k] = (e0 + k) * 2;
    }
  }
  if constexpr (NORM) {
    if (p.w_lds) {
      __builtin_amdgcn_s_waitcnt(0x0070);                   // vmcnt(0): the weight vector (and row 0) have landed
      __syncthreads();
#pragma unroll
      for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int k = 0; k < 16; ++k) wg[ps][k] = *reinterpret_cast<const half_t *>(wbuf + off[ps][k]);
    } else {                                                // (hidden 16384 with the fused residual add: LDS is full)
#pragma unroll
      for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int k = 0; k < 16; ++k) wg[ps][k] = p.b[off[ps][k] >> 1];
    }
  }

  unsigned wp[NP][8];                                       // SIM, FP16-domain form: the weights as pairs (pair_lo(i), + 4)
  if constexpr (NORM && SIM) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        wp[ps][i] = (unsigned)__builtin_bit_cast(unsigned short, wg[ps][pair_lo(i)]) |
                    ((unsigned)__builtin_bit_cast(unsigned short, wg[ps][pair_lo(i) + 4]) << 16);
  }

  // (Round 4 tried issuing a row's stores one iteration late, behind the next row's vmcnt(0) -- vmcnt counts stores too, so that wait
  // also covers the previous row's writes: slower, 14.6 vs 13.6 us at 4,096 x 4096; the rows of the six resident workgroups of a CU
  // already overlap each other's write latency, and the 10-25 registers of the parked results cost a workgroup per CU.)
  SlotOut<DQ> pend[NP];
  auto flush = [&](int64_t rr) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int slot = ps * 256 + tid;
      if (slot < nslots) {
        const int g = slot >> 3;
        slot_store<DQ, FMT>(pend[ps], p, rr, g, j, slot * 16, g == Gt - 1, K4h);
      }
    }
  };
  // one row: bc = the buffer it sits in as a compile-time constant (HC instances: 0 / 1) or -1 = the run-time value brt
  auto do_row = [&](auto bc, const int64_t r, const int brt) {
    constexpr int BC = decltype(bc)::value;
    const int b = BC < 0 ? brt : BC;
    __builtin_amdgcn_s_waitcnt(0x0070);                     // vmcnt(0): my DMA writes have landed
    __syncthreads();
    const int64_t rn = r + rstep;
    if (rn < rend) issue_row(rn, b ^ 1);
    char *row = smem + b * stage;

    // Sum of squares: a FIXED-SHAPE FP32 tree over the row in memory order (not over the gathered channels, so the reorder index
    // does not enter): 16-byte chunk c belongs to thread (wave (c / 64) % 4, lane c % 64); a thread folds its chunks in order with
    // s = fma(x, x, s); lanes combine by the butterfly xor 32, 16, .., 1; the four waves as ((w0 + w1) + w2) + w3.  Deterministic,
    // restated step by step in oracle/ (r01 used an FP64 sum rounded once: two half-rate instructions per element).
    float ss = 0.f;
    if constexpr (ADD) {
      // x + residual (one fp16 add per element, as torch's half add), written back to the residual stream and kept in
      // LDS for the gather; the sum of squares is taken here, on the linear data
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int i = 0; i < 2 * NP; ++i) {
        const int c = (i * 4 + wave) * 64 + lane;
        if (c < nchunks) {
          const h8 a = *reinterpret_cast<const h8 *>(row + c * 16);
          const h8 rr = *reinterpret_cast<const h8 *>(row + bufbytes + c * 16);
          const h8 sum = a + rr;
          *reinterpret_cast<h8 *>(row + c * 16) = sum;
          *reinterpret_cast<h8 *>(reinterpret_cast<char *>(p.res_out + r * (int64_t)H) + c * 16) = sum;
#pragma unroll
          for (int k = 0; k < 8; ++k) ss = __builtin_fmaf((float)sum[k], (float)sum[k], ss);
        }
      }
      ss = wave_sum_butterfly(ss);
      if (lane == 0) red[b * 4 + wave] = ss;
      __syncthreads();                                      // sums visible to the gather, partial sums to everyone
    } else if constexpr (NORM) {
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int i = 0; i < 2 * NP; ++i) {
        const int c = (i * 4 + wave) * 64 + lane;
        if (c < nchunks) {
          const h8 a = *reinterpret_cast<const h8 *>(row + c * 16);
#pragma unroll
          for (int k = 0; k < 8; ++k) ss = __builtin_fmaf((float)a[k], (float)a[k], ss);
        }
      }
      ss = wave_sum_butterfly(ss);
      if (lane == 0) red[b * 4 + wave] = ss;
      __syncthreads();
    }

    float rinv = 0.f;
    if constexpr (NORM) {
      const float tot = ((red[b * 4 + 0] + red[b * 4 + 1]) + red[b * 4 + 2]) + red[b * 4 + 3];
      // correctly rounded divide, sqrt and divide (hipcc default); a power-of-two H divides exactly by multiplying
      const float var = (H & (H - 1)) == 0 ? tot * (1.0f / (float)H) : tot / (float)H;
      rinv = rinv_sqrt_exact(var + p.eps);
    }
    if constexpr (SIM) {
      // the simulated path in the FP16 domain (quant_math.h): a slot as 8 half pairs, pair i = channels (pair_lo(i), pair_lo(i) + 4)
      if (p.clip >= kSimHalfMinClip) {
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
          unsigned z[8];
          if (p.idx) {
            unsigned xs[16];                              // one gathered half per register (ds_read_u16 zero-extends)
#pragma unroll
            for (int k = 0; k < 16; ++k) xs[k] = *reinterpret_cast<const unsigned short *>(row + off[ps][k]);
            if constexpr (NORM) {
              // HF LlamaRMSNorm, half opmath: half(x * rinv) from the FP32 product, then a half multiply by the weight
              sim_scale4<0, 0, 0, 0>(xs[0], xs[1], xs[2], xs[3], xs[4], xs[5], xs[6], xs[7], rinv, z[0], z[1], z[2], z[3]);
              sim_scale4<0, 0, 0, 0>(xs[8], xs[9], xs[10], xs[11], xs[12], xs[13], xs[14], xs[15], rinv, z[4], z[5], z[6], z[7]);
            } else {
#pragma unroll
              for (int i = 0; i < 8; ++i) z[i] = xs[pair_lo(i)] | (xs[pair_lo(i) + 4] << 16);
            }
          } else {
            v4u raw[2];                                   // dword m = channels (2m, 2m + 1)
            raw[0] = *reinterpret_cast<const v4u *>(row + off[ps][0]);
            raw[1] = *reinterpret_cast<const v4u *>(row + off[ps][0] + 16);
            if constexpr (NORM) {
              sim_scale4<0, 1, 0, 1>(raw[0][0], raw[0][0], raw[0][1], raw[0][1], raw[0][2], raw[0][2], raw[0][3], raw[0][3], rinv, z[0],
                                     z[1], z[2], z[3]);
              sim_scale4<0, 1, 0, 1>(raw[1][0], raw[1][0], raw[1][1], raw[1][1], raw[1][2], raw[1][2], raw[1][3], raw[1][3], rinv, z[4],
                                     z[5], z[6], z[7]);
            } else {
#pragma unroll
              for (int k = 0; k < 2; ++k) {
                z[4 * k + 0] = __builtin_amdgcn_perm(raw[k][2], raw[k][0], 0x05040100u);   // channels 8k + (0, 4)
                z[4 * k + 1] = __builtin_amdgcn_perm(raw[k][2], raw[k][0], 0x07060302u);   // (1, 5)
                z[4 * k + 2] = __builtin_amdgcn_perm(raw[k][3], raw[k][1], 0x05040100u);   // (2, 6)
                z[4 * k + 3] = __builtin_amdgcn_perm(raw[k][3], raw[k][1], 0x07060302u);   // (3, 7)
              }
            }
          }
          if constexpr (NORM) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
              z[i] = __builtin_bit_cast(unsigned, __builtin_bit_cast(h2v, z[i]) * __builtin_bit_cast(h2v, wp[ps][i]));
          }
          const int slot = ps * 256 + tid;
          if (slot < nslots) {
            const int g = slot >> 3;
            pend[ps] = slot_codes_h<DQ, FMT>(z, p, g == Gt - 1);
          }
        }
        flush(r);
        return;
      }
    }

    float x[NP][16];
    half_t xh[NP][16];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      if (p.idx) {
#pragma unroll
        for (int k = 0; k < 16; ++k) xh[ps][k] = *reinterpret_cast<const half_t *>(row + off[ps][k]);
      } else {
        v4u raw[2];
        raw[0] = *reinterpret_cast<const v4u *>(row + off[ps][0]);
        raw[1] = *reinterpret_cast<const v4u *>(row + off[ps][0] + 16);
        const half_t *hv = reinterpret_cast<const half_t *>(raw);
#pragma unroll
        for (int k = 0; k < 16; ++k) xh[ps][k] = hv[k];
      }
    }
    if constexpr (NORM) {
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if constexpr (SIM) {
            // HF LlamaRMSNorm, half opmath: half(x * rinv) from the FP32 product, then a half multiply by the weight (the exact
            // product of two halves rounded once: what round_h(w * y) in FP32 gives)
            const half_t y = (half_t)((float)xh[ps][k] * rinv);
            x[ps][k] = (float)(half_t)(wg[ps][k] * y);
          } else {
            x[ps][k] = (float)(half_t)(((float)xh[ps][k] * (float)wg[ps][k]) * rinv);   // RMSNorm.cuh:145-151
          }
        }
      }
    } else {
#pragma unroll
      for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int k = 0; k < 16; ++k) x[ps][k] = (float)xh[ps][k];
    }
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int slot = ps * 256 + tid;
      if (slot < nslots) {
        const int g = slot >> 3;
        pend[ps] = slot_codes<SIM, DQ, FMT>(x[ps], p, g == Gt - 1);
      }
    }
    flush(r);
  };
  if constexpr (HC != 0) {
    while (true) {
      if (r >= rend) break;
      do_row(std::integral_constant<int, 0>(), r, 0);
      r += rstep;
      if (r >= rend) break;
      do_row(std::integral_constant<int, 1>(), r, 1);
      r += rstep;
    }
  } else {
    for (; r < rend; r += rstep, b ^= 1) do_row(std::integral_constant<int, -1>(), r, b);
  }
}

// silu(a)*b: no gather, no LDS; one row per workgroup.
template <bool SIM, bool DQ, int FMT>
__global__ __launch_bounds__(256) void silu_quant2_kernel(ActQuantParams p) {
  const int tid = threadIdx.x;
  {  // the kernel arguments in ONE batch of scalar loads (round 6: hipcc fetched them in two or three, a scalar-cache round trip apart;
     // at decode batches this kernel is a chain of such trips)
    const void *a0 = p.x, *a1 = p.b, *a2 = p.res, *a3 = p.res_out, *a4 = p.idx, *a5 = p.o8, *a6 = p.o4, *a7 = p.s8, *a8_ = p.s4, *a9 = p.xq;
    const int i0 = (int)p.M, i1 = p.H, i2 = p.ref_layout, i3 = p.w_lds, i4 = (int)gridDim.x, i5 = (int)p.f6_rows, i6 = (int)p.ld;
    asm volatile("" ::"s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(a6), "s"(a7), "s"(a8_), "s"(a9), "s"(i0), "s"(i1), "s"(i2), "s"(i3), "s"(i4), "s"(i5), "s"(i6));
  }
  // rows by XCD (see act_quant2_kernel): the grid is 8 * ceil(M / 8); XCD x takes rows [x * cm, (x + 1) * cm)
  const int64_t cm = (p.M + 7) >> 3;
  const int64_t r = (blockIdx.x & 7) * cm + (blockIdx.x >> 3);
  if (r >= p.M) return;
  const int H = p.H;
  const int nslots = H >> 4;
  const int Gt = H >> 7;
  const int K4h = (H - kKeeper) >> 1;
  const half_t *arow = p.x + r * (int64_t)H, *brow = p.b + r * (int64_t)H;
  // gridDim.y > 1 (decode batches): a row is shared by gridDim.y workgroups, 256 slots each -- the 128-channel groups are independent
  // (H = 11008 at 1..16 rows: 4.5 -> 2.7 us under graph replay, tools/r02/quant_latency_probe.py)
  for (int slot = blockIdx.y * 256 + tid; slot < nslots; slot += 256 * gridDim.y) {
    const int e0 = slot * 16;
    v4u ra[2], rb[2];
    ra[0] = *reinterpret_cast<const v4u *>(arow + e0);
    ra[1] = *reinterpret_cast<const v4u *>(arow + e0 + 8);
    rb[0] = *reinterpret_cast<const v4u *>(brow + e0);
    rb[1] = *reinterpret_cast<const v4u *>(brow + e0 + 8);
    const half_t *av = reinterpret_cast<const half_t *>(ra);
    const half_t *bv = reinterpret_cast<const half_t *>(rb);
    const int g = slot >> 3;
    if constexpr (SIM) {
      if (p.clip >= kSimHalfMinClip) {
        // act_fn(gate) * up in half opmath: half(silu) from the FP32 value, then a half multiply -- pairwise (quant_math.h)
        unsigned z[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int L = pair_lo(i);
          const h2v sh = {(_Float16)silu_f32((float)av[L]), (_Float16)silu_f32((float)av[L + 4])};
          const h2v bb = {bv[L], bv[L + 4]};
          z[i] = __builtin_bit_cast(unsigned, sh * bb);
        }
        quant_slot_h<DQ, FMT>(z, p, r, g, tid & 7, e0, g == Gt - 1, K4h);
        continue;
      }
    }
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      v[k] = silu_mul<SIM>((float)av[k], (float)bv[k]);
    }
    quant_slot<SIM, DQ, FMT>(v, p, r, g, tid & 7, e0, g == Gt - 1, K4h);
  }
}

template <class K>
static int resident_blocks(K kernel, size_t lds) {           // persistent grid: workgroups the chip holds at once
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    cus = 256;
  return per_cu * cus;
}

template <int OP, bool SIM, bool DQ, int NP, int FMT, int HC = 0>
static int launch_act_quant2_np(const ActQuantParams &p0, hipStream_t s) {
  ActQuantParams p = p0;
  const size_t rowb = (size_t)((p.H * 2 + 1023) & ~1023);
  constexpr bool NORMOP = OP == OP_RMSNORM || OP == OP_ADD_RMSNORM;
  const size_t base = (OP == OP_ADD_RMSNORM ? 4 : 2) * rowb + 1024;
  p.w_lds = NORMOP && base + rowb <= 160 * 1024;
  size_t lds = base + (p.w_lds ? rowb : 0);
#ifdef ATOM_TOOLS   // occupancy experiments: unused LDS that lowers the number of resident workgroups (profiles/r05/quant_occupancy.txt)
  if (const char *e = getenv("ATOM_Q_EXTRA_LDS")) lds = lds + (size_t)atoi(e) <= 160 * 1024 ? lds + (size_t)atoi(e) : lds;
#endif
  // the dynamic-LDS attribute is per device (ensure_max_lds); the occupancy answer depends on the device and on the LDS size, i.e.
  // on H: one cached (lds, resident) word per device slot and kernel (a race re-computes the same value)
  static std::atomic<uint64_t> lds_set{0};
  static std::atomic<uint64_t> cache[64];
  const auto kernel = act_quant2_kernel<OP, SIM, DQ, NP, FMT, HC>;
  if (const int st = ensure_max_lds(reinterpret_cast<const void *>(kernel), 160 * 1024, lds_set); st != ATOM_OK) return st;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ATOM_ERR_LAUNCH;
  std::atomic<uint64_t> &slot = cache[dev & 63];
  uint64_t c = slot.load(std::memory_order_relaxed);
  if ((c >> 32) != lds) {
    c = ((uint64_t)lds << 32) | (uint32_t)resident_blocks(kernel, lds);
    slot.store(c, std::memory_order_relaxed);
  }
  const int resident = (int)(uint32_t)c;
  const unsigned grid = (unsigned)(p.M < resident ? p.M : resident);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, s, p);
  return ATOM_OK;
}

template <int OP, bool SIM, bool DQ, int FMT>
static int launch_act_quant2_fmt(const ActQuantParams &p, hipStream_t s) {
  if constexpr (OP == OP_SILU_MUL) {
    const unsigned parts = p.M <= 1024 ? (unsigned)(((p.H >> 4) + 255) >> 8) : 1u;
    hipLaunchKernelGGL((silu_quant2_kernel<SIM, DQ, FMT>), dim3((unsigned)(((p.M + 7) >> 3) << 3), parts), dim3(256), 0, s, p);
    return ATOM_OK;
  } else {
    const int np = ((p.H >> 4) + 255) >> 8;
    // the hidden size of Llama-7B has its own instances of the RMSNorm kernels (same box, 4096 / 65,536 rows: RMSNorm-quant 14.2-14.7 ->
    // 13.7-13.8 / 160-162 -> 153-156 us, kernel-flavoured 15.0-15.8 -> 14.1-14.2 / 166 -> 158; the plain reorder gains nothing at
    // 4,096 rows and loses 9 % at 65,536 in the kernel-flavoured mode: it keeps the generic form; profiles/r05/quant_valu.txt)
    if constexpr (OP != OP_REORDER) {
      if (np == 1 && p.H == 4096) return launch_act_quant2_np<OP, SIM, DQ, 1, FMT, 4096>(p, s);
    }
    if (np == 1) return launch_act_quant2_np<OP, SIM, DQ, 1, FMT>(p, s);
    if (np == 2) return launch_act_quant2_np<OP, SIM, DQ, 2, FMT>(p, s);
    if (np == 3) return launch_act_quant2_np<OP, SIM, DQ, 3, FMT>(p, s);
    return launch_act_quant2_np<OP, SIM, DQ, 4, FMT>(p, s);
  }
}
template <int OP, bool SIM, bool DQ>
static int launch_act_quant2(const ActQuantParams &p, hipStream_t s) {
  if (p.f6_rows) return launch_act_quant2_fmt<OP, SIM, DQ, 2>(p, s);
  if (p.wide) return launch_act_quant2_fmt<OP, SIM, DQ, 1>(p, s);
  return launch_act_quant2_fmt<OP, SIM, DQ, 0>(p, s);
}

static int launch_act_quant(int op, ActQuantParams p, int quant_mode, int scale_layout, void *stream) {
  if (!p.x || !p.o8 || !p.o4 || !p.s8 || !p.s4) return ATOM_ERR_INVALID_ARG;
  if (op != OP_REORDER && !p.b) return ATOM_ERR_INVALID_ARG;
  if (op == OP_ADD_RMSNORM && (!p.res || !p.res_out)) return ATOM_ERR_INVALID_ARG;
  if (op == OP_ADD_RMSNORM && (!aligned16(p.res) || !aligned16(p.res_out))) return ATOM_ERR_ALIGN;
  p.wide = (quant_mode & ATOM_QUANT_WIDE_CODES) != 0;
  p.f6_rows = (quant_mode & ATOM_QUANT_F6_CODES) ? (p.M + 255) / 256 * 256 : 0;
  if (p.wide && p.f6_rows) return ATOM_ERR_INVALID_ARG;
  quant_mode &= ~(ATOM_QUANT_WIDE_CODES | ATOM_QUANT_F6_CODES);
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
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int st = ATOM_OK;
#define ATOM_LAUNCH2(OPV)                                                  \
  if (p.sim && p.xq) st = launch_act_quant2<OPV, true, true>(p, s);        \
  else if (p.sim) st = launch_act_quant2<OPV, true, false>(p, s);          \
  else if (p.xq) st = launch_act_quant2<OPV, false, true>(p, s);           \
  else st = launch_act_quant2<OPV, false, false>(p, s);
  switch (op) {
    case OP_REORDER: ATOM_LAUNCH2(OP_REORDER) break;
    case OP_RMSNORM: ATOM_LAUNCH2(OP_RMSNORM) break;
    case OP_ADD_RMSNORM: ATOM_LAUNCH2(OP_ADD_RMSNORM) break;
    default: ATOM_LAUNCH2(OP_SILU_MUL) break;
  }
#undef ATOM_LAUNCH2
  return st != ATOM_OK ? st : check_launch();
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

// ------------------------------------------------------------------------------------------------
// KV-cache fake quantisation of the simulated path: asymmetric n-bit per 128-d head vector, FP16 opmath
//   reference: quantize_attn_k_wrapper / quantize_attn_v_wrapper (model/quant.py:233-257) -> quantize_tensor(sym=False)
//   (model/quant.py:143-145,173-181).  8 lanes per vector (16 elements each), max / min by DPP, the same exact
//   3-FMA quotient as the activation quantisers.  Input may be a strided [B, H, S, 128] view (the transposed projection
//   output), output is contiguous -- what the reference's reshape(-1, 128) ... view(saved_shape) produces.
struct KvFqParams {
  const half_t *x;
  half_t *y;
  int64_t nvec;
  int H, S;
  int64_t sb, sh, ss;    // element strides of x over (batch, head, position); the last dim is contiguous
  float qmax, clip;
};

__device__ __forceinline__ float min8(float a) {
  a = fminf(a, dpp_f<0xB1>(a));
  a = fminf(a, dpp_f<0x4E>(a));
  a = fminf(a, dpp_f<0x141>(a));
  return a;
}

__device__ __forceinline__ float exact_div_h(float n, float d, float rd) {   // RN_f32(n / d), rd = RN(1/d); fp16-valued n, d
  const float q0 = n * rd;
  return __builtin_fmaf(__builtin_fmaf(-q0, d, n), rd, q0);
}

__global__ __launch_bounds__(256) void kv_fake_quant_kernel(KvFqParams p) {
  const int64_t v = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int j = threadIdx.x & 7;
  const int64_t vc = v < p.nvec ? v : p.nvec - 1;                 // keep all lanes alive for the DPP reductions
  const int64_t b = vc / ((int64_t)p.H * p.S), h = (vc / p.S) % p.H, sq = vc % p.S;
  const half_t *src = p.x + b * p.sb + h * p.sh + sq * p.ss + j * 16;
  v4u raw[2];
  raw[0] = *reinterpret_cast<const v4u *>(src);
  raw[1] = *reinterpret_cast<const v4u *>(src + 8);
  const half_t *hv = reinterpret_cast<const half_t *>(raw);
  float x[16];
  float mx = -INFINITY, mn = INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    x[i] = (float)hv[i];
    mx = fmaxf(mx, x[i]);
    mn = fminf(mn, x[i]);
  }
  mx = max8(mx);
  mn = min8(mn);
  if (p.clip < 1.0f) {                                            // quant.py:176-178 (in-place half multiplies)
    mx = round_h(mx * p.clip);
    mn = round_h(mn * p.clip);
  }
  const float range = fmaxf(round_h(mx - mn), (float)(half_t)1e-5f);          // (w_max - w_min).clamp(min=1e-5)
  const float s = round_h(exact_div_h(opaque(range), p.qmax, 1.0f / p.qmax)); // / q_max (qmax = 2^n - 1 is an fp16 value)
  const float so = opaque(s);
  const float r0 = __builtin_amdgcn_rcpf(so);
  const float rs = __builtin_fmaf(__builtin_fmaf(-r0, so, 1.0f), r0, r0);     // RN(1/s), round_probe.cpp
  const float base = __builtin_amdgcn_fmed3f(rintf(round_h(exact_div_h(-mn, so, rs))), 0.f, p.qmax);   // :180
  v4u o[2];
  half_t *ov = reinterpret_cast<half_t *>(o);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    // clamp(round(w / scales) + base, 0, q_max) - base: the half additions are exact below 2048 and saturate above
    const float c = __builtin_amdgcn_fmed3f(rintf(round_h(exact_div_h(x[i], so, rs))) + base, 0.f, p.qmax);
    ov[i] = (half_t)__builtin_fmaf(c - base, s, 0.0f);           // exact product, one rounding
  }
  if (v < p.nvec) {
    v4u *dst = reinterpret_cast<v4u *>(p.y + v * 128 + j * 16);
    dst[0] = o[0];
    dst[1] = o[1];
  }
}

// ------------------------------------------------------------------------------------------------
// Packed INT4 weights -> the F6 operand format of gemm_w4a4_f6.hip ([G][rows_pad][104] BF6 streams).  Offline.
struct RepackF6Params {
  const uint8_t *B4;
  uint8_t *out;
  int64_t N, rows_pad;
  int K4h, G;
  const half_t *scale;     // optional: per-(row, group) scales to embed at byte 96 (activation operands); NULL for weights
  int64_t ld;              // halves between groups of `scale`
  int ref_layout;
  const half_t *wscale;    // optional (weights): fp16 scales [G, N] ...
  float *wscale_out;       // ... written as float32 [G][rows_pad] (ATOM_SB_F32)
};

// One workgroup per (group, block of 256 rows): thread t re-codes the 128 codes of row t (64 packed bytes -> 96 bytes of
// BF6 fields + the scale) into LDS in record order; the block's 26,624 output bytes are contiguous in the group-major
// format and leave as 16 bytes per lane.  (The first version wrote 12 bytes per thread at a 12-byte stride straight to
// HBM: 2.5 TB/s; this one is bound by the 64-byte row pieces it reads.)
struct RepackF6Pair { RepackF6Params op[2]; };

__global__ __launch_bounds__(256) void repack_f6_kernel(RepackF6Pair pp) {
  const RepackF6Params &p = pp.op[blockIdx.z];
  if ((int64_t)blockIdx.x * 256 >= p.rows_pad) return;
  typedef float v16f __attribute__((ext_vector_type(16)));
  typedef unsigned v6u __attribute__((ext_vector_type(6)));
  __shared__ __attribute__((aligned(16))) unsigned char rec[256 * 104];
  const int g = blockIdx.y;
  const int64_t n0 = (int64_t)blockIdx.x * 256;
  const int t = threadIdx.x;
  const int64_t n = n0 + t;
  unsigned *dst = reinterpret_cast<unsigned *>(rec + t * 104);
  if (n < p.N) {
    const uint8_t *src = p.B4 + n * p.K4h + g * 64;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                          // 32 codes = 16 packed bytes -> 24 bytes
      const v4u raw = *reinterpret_cast<const v4u *>(src + 16 * q);
      v16f ea, eb;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const unsigned byte = (raw[i >> 2] >> (8 * (i & 3))) & 0xFF;
        ea[i] = (float)((int)(byte << 28) >> 28);          // element 2i: low nibble, sign-extended
        eb[i] = (float)((int)(byte << 24) >> 28);          // element 2i+1: high nibble
      }
      // (the builtin, not cvt_2xpk16_bf6: see common.h -- guarded by the disassembly check in tests/test_abi_cpu.py)
      const v6u f = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(ea, eb, 1.0f);
#pragma unroll
      for (int k = 0; k < 6; ++k) dst[6 * q + k] = f[k];
    }
    unsigned sc = 0u, sc32 = 0u;
    if (p.scale) {
      const half_t sv = p.scale[(int64_t)g * p.ld + (p.ref_layout ? ref_scale_index((int)n) : (int)n)];
      sc = (unsigned)__builtin_bit_cast(unsigned short, sv);
      sc32 = __builtin_bit_cast(unsigned, (float)sv);
    }
    dst[24] = sc;
    dst[25] = sc32;
  } else {
#pragma unroll
    for (int k = 0; k < 26; ++k) dst[k] = 0u;              // pad rows: zeros
  }
  if (p.wscale_out) p.wscale_out[(int64_t)g * p.rows_pad + n] = n < p.N ? (float)p.wscale[(int64_t)g * p.N + n] : 0.f;
  __syncthreads();
  uint8_t *out = p.out + ((int64_t)g * p.rows_pad + n0) * 104;
  for (int i = t; i < 256 * 104 / 16; i += 256)
    *reinterpret_cast<v4u *>(out + i * 16) = *reinterpret_cast<const v4u *>(rec + i * 16);
}

// The same for a FEW rows (activations of mid-size batches: the re-coding launch in front of the mid-size-batch GEMM): a block of 64
// rows x one group per workgroup, four threads per row (32 codes = one conversion each) -- 4 x (rows / 64) x G waves instead of
// (rows / 256) x G workgroups of one row per thread, which is 31 workgroups on a 256-CU chip at 256 x 4096.  Rows [rows, 64-row block
// end) are zero records, rows beyond are not written (pad rows of an F6 operand may hold any bytes, include/atom_hip.h).
__global__ __launch_bounds__(256) void repack_f6_rows64_kernel(RepackF6Params p) {
  typedef float v16f __attribute__((ext_vector_type(16)));
  typedef unsigned v6u __attribute__((ext_vector_type(6)));
  __shared__ __attribute__((aligned(16))) unsigned char rec[64 * 104];
  const int g = blockIdx.y;
  const int64_t n0 = (int64_t)blockIdx.x * 64;
  const int t = threadIdx.x, r = t >> 2, q = t & 3;
  const int64_t n = n0 + r;
  unsigned *dst = reinterpret_cast<unsigned *>(rec + r * 104);
  if (n < p.N) {
    const v4u raw = *reinterpret_cast<const v4u *>(p.B4 + n * p.K4h + g * 64 + 16 * q);
    v16f ea, eb;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const unsigned byte = (raw[i >> 2] >> (8 * (i & 3))) & 0xFF;
      ea[i] = (float)((int)(byte << 28) >> 28);
      eb[i] = (float)((int)(byte << 24) >> 28);
    }
    const v6u f = cvt_2xpk16_bf6(ea, eb);
#pragma unroll
    for (int k = 0; k < 6; ++k) dst[6 * q + k] = f[k];
    if (q == 0) {
      const half_t sv = p.scale[(int64_t)g * p.ld + (p.ref_layout ? ref_scale_index((int)n) : (int)n)];
      dst[24] = (unsigned)__builtin_bit_cast(unsigned short, sv);
      dst[25] = __builtin_bit_cast(unsigned, (float)sv);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) dst[6 * q + k] = 0u;
    if (q == 0) { dst[24] = 0u; dst[25] = 0u; }
  }
  __syncthreads();
  uint8_t *out = p.out + ((int64_t)g * p.rows_pad + n0) * 104;
  for (int i = t; i < 64 * 104 / 16; i += 256)
    *reinterpret_cast<v4u *>(out + i * 16) = *reinterpret_cast<const v4u *>(rec + i * 16);
}

// packed operand [rows, K4/2] (+ its scales, for activations) -> F6 buffer [G][round_up(rows, 256)][104]
int launch_repack_f6(const uint8_t *src, int64_t rows, int K4h, int G, const half_t *scale, int64_t ld, int ref_layout,
                     uint8_t *out, hipStream_t s) {
  if (scale && rows <= 2048) {                             // few rows: 64-row blocks, four threads per row
    const RepackF6Params q{src, out, rows, (rows + 255) / 256 * 256, K4h, G, scale, ld, ref_layout, nullptr, nullptr};
    hipLaunchKernelGGL(repack_f6_rows64_kernel, dim3((unsigned)((rows + 63) / 64), (unsigned)G), dim3(256), 0, s, q);
    return check_launch();
  }
  RepackF6Pair pp;
  pp.op[0] = RepackF6Params{src, out, rows, (rows + 255) / 256 * 256, K4h, G, scale, ld, ref_layout, nullptr, nullptr};
  pp.op[1] = pp.op[0];
  hipLaunchKernelGGL(repack_f6_kernel, dim3((unsigned)(pp.op[0].rows_pad / 256), (unsigned)G, 1), dim3(256), 0, s, pp);
  return check_launch();
}

// both operands of a GEMM in one launch: activations (with their scales) and weights
int launch_repack_f6_pair(const uint8_t *A4, int64_t M, const half_t *sA, int64_t ldA, int ref_layout, uint8_t *outA,
                          const half_t *sB, float *sB32,
                          const uint8_t *B4, int64_t N, uint8_t *outB, int K4h, int G, hipStream_t s) {
  RepackF6Pair pp;
  pp.op[0] = RepackF6Params{A4, outA, M, (M + 255) / 256 * 256, K4h, G, sA, ldA, ref_layout, nullptr, nullptr};
  pp.op[1] = RepackF6Params{B4, outB, N, (N + 255) / 256 * 256, K4h, G, nullptr, 0, 0, sB, sB32};
  const int64_t rp = pp.op[0].rows_pad > pp.op[1].rows_pad ? pp.op[0].rows_pad : pp.op[1].rows_pad;
  hipLaunchKernelGGL(repack_f6_kernel, dim3((unsigned)(rp / 256), (unsigned)G, 2), dim3(256), 0, s, pp);
  return check_launch();
}

// ------------------------------------------------------------------------------------------------
// Checkers for the two caller ASSERTIONS of the GEMM entry points (ATOM_B_SCALE_PAIRS, ATOM_WS_WEIGHT_CACHED): a wrong assertion gives
// wrong numbers, never a fault, so a binding that cannot prove them by construction asks these (offline with the weight, or through
// ATOM_WS_VERIFY on a debug call).  Both add the number of violations to a device counter.
__global__ __launch_bounds__(256) void check_scale_pairs_kernel(const half_t *sB, int64_t npairs_total, int32_t *n_bad) {
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npairs_total; i += (int64_t)gridDim.x * 256) {
    const unsigned w = reinterpret_cast<const unsigned *>(sB)[i];           // channels 2 j, 2 j + 1 of one group: N is even, sB 4-byte aligned
    bad += (w & 0xFFFFu) != (w >> 16);
  }
  if (bad) atomicAdd(n_bad, bad);
}

// the F6 weight form [G][rows_pad][104] (+ float32 scales [G][rows_pad] behind it) against what atom_repack_weight_f6s makes of the
// packed weight: one thread per (row < N, group) re-codes its 128 codes and compares the record and the scale (pad rows: not compared)
__global__ __launch_bounds__(256) void verify_weight_f6s_kernel(RepackF6Params p, const uint8_t *have, int32_t *n_bad) {
  typedef float v16f __attribute__((ext_vector_type(16)));
  typedef unsigned v6u __attribute__((ext_vector_type(6)));
  const int g = blockIdx.y;
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= p.N) return;
  const uint8_t *src = p.B4 + n * p.K4h + g * 64;
  const unsigned *rec = reinterpret_cast<const unsigned *>(have + ((int64_t)g * p.rows_pad + n) * 104);
  int bad = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const v4u raw = *reinterpret_cast<const v4u *>(src + 16 * q);
    v16f ea, eb;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const unsigned byte = (raw[i >> 2] >> (8 * (i & 3))) & 0xFF;
      ea[i] = (float)((int)(byte << 28) >> 28);
      eb[i] = (float)((int)(byte << 24) >> 28);
    }
    const v6u f = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(ea, eb, 1.0f);
#pragma unroll
    for (int k = 0; k < 6; ++k) bad |= rec[6 * q + k] != f[k];
  }
  const float *s32 = reinterpret_cast<const float *>(have + (size_t)p.G * p.rows_pad * 104);
  const float want = (float)p.wscale[(int64_t)g * p.N + n];
  bad |= __builtin_bit_cast(unsigned, s32[(int64_t)g * p.rows_pad + n]) != __builtin_bit_cast(unsigned, want);
  if (bad) atomicAdd(n_bad, 1);
}

int launch_check_scale_pairs(const half_t *sB, int64_t G, int64_t N, int32_t *n_bad, hipStream_t s) {
  const int64_t np = G * (N / 2);
  const unsigned grid = (unsigned)(np / 256 + 1 < 1024 ? np / 256 + 1 : 1024);
  hipLaunchKernelGGL(check_scale_pairs_kernel, dim3(grid), dim3(256), 0, s, sB, np, n_bad);
  return check_launch();
}

int launch_verify_weight_f6s(const uint8_t *B4, const half_t *sB, int64_t N, int K4h, int G, const uint8_t *have, int32_t *n_bad, hipStream_t s) {
  const int64_t Npad = (N + 255) / 256 * 256;
  const RepackF6Params q{B4, nullptr, N, Npad, K4h, G, nullptr, 0, 0, sB, nullptr};
  hipLaunchKernelGGL(verify_weight_f6s_kernel, dim3((unsigned)(Npad / 256), (unsigned)G), dim3(256), 0, s, q, have, n_bad);
  return check_launch();
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

int atom_add_rmsnorm_reorder_quant_f16(const void *x, const void *residual, void *residual_out, const void *weight,
                                       float eps, const int16_t *reorder_index, int64_t M, int hidden, int quant_mode, float clip,
                                       int scale_layout, void *o_outliers, void *o_norms, void *outlier_scales,
                                       void *norm_scales, void *xq_f16, void *stream) {
  ActQuantParams p{};
  p.x = (const half_t *)x; p.res = (const half_t *)residual; p.res_out = (half_t *)residual_out;
  p.b = (const half_t *)weight; p.eps = eps;
  p.idx = reorder_index; p.M = M; p.H = hidden; p.clip = clip;
  p.o8 = (int8_t *)o_outliers; p.o4 = (uint8_t *)o_norms; p.s8 = (half_t *)outlier_scales;
  p.s4 = (half_t *)norm_scales; p.xq = (half_t *)xq_f16;
  return launch_act_quant(OP_ADD_RMSNORM, p, quant_mode, scale_layout, stream);
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

int atom_kv_fake_quant_f16(const void *x, void *y, int64_t batch, int num_heads, int64_t seq_len, int64_t stride_b,
                           int64_t stride_h, int64_t stride_s, int n_bits, float clip, void *stream) {
  if (!x || !y) return ATOM_ERR_INVALID_ARG;
  if (n_bits < 2 || n_bits > 8 || !(clip > 0.f)) return ATOM_ERR_INVALID_ARG;
  if (batch < 1 || num_heads < 1 || seq_len < 1 || batch * num_heads * seq_len > (int64_t(1) << 40)) return ATOM_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(y) || (stride_b % 8) || (stride_h % 8) || (stride_s % 8)) return ATOM_ERR_ALIGN;
  KvFqParams p{(const half_t *)x, (half_t *)y, batch * num_heads * seq_len, num_heads, (int)seq_len, stride_b, stride_h,
               stride_s, (float)((1 << n_bits) - 1), clip};
  if (seq_len > 0x7fffffff) return ATOM_ERR_SHAPE;
  hipLaunchKernelGGL(kv_fake_quant_kernel, dim3((unsigned)((p.nvec + 31) / 32)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), p);
  return check_launch();
}

size_t atom_f6_rows(int64_t rows) { return rows < 0 ? 0 : (size_t)((rows + 255) / 256 * 256); }

int atom_repack_weight_f6(const void *B4, int64_t N, int64_t K_total, void *B_f6, void *stream) {
  if (!B4 || !B_f6) return ATOM_ERR_INVALID_ARG;
  if (N < 1 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0 || K_total > (1 << 20)) return ATOM_ERR_SHAPE;
  if (!aligned16(B4) || !aligned16(B_f6)) return ATOM_ERR_ALIGN;
  return launch_repack_f6((const uint8_t *)B4, N, (int)((K_total - kKeeper) / 2), (int)((K_total - kKeeper) / kGroup), nullptr, 0,
                          0, (uint8_t *)B_f6, reinterpret_cast<hipStream_t>(stream));
}

int atom_repack_act_f6(const void *A4, const void *sA, int64_t M, int64_t K_total, int scale_layout, void *A_f6, void *stream) {
  if (!A4 || !sA || !A_f6) return ATOM_ERR_INVALID_ARG;
  if (scale_layout != ATOM_SCALE_LAYOUT_REF && scale_layout != ATOM_SCALE_LAYOUT_PLAIN) return ATOM_ERR_INVALID_ARG;
  if (M < 1 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0 || K_total > (1 << 20)) return ATOM_ERR_SHAPE;
  if (!aligned16(A4) || !aligned16(A_f6)) return ATOM_ERR_ALIGN;
  return launch_repack_f6((const uint8_t *)A4, M, (int)((K_total - kKeeper) / 2), (int)((K_total - kKeeper) / kGroup), (const half_t *)sA,
                          (int64_t)atom_scale_size(M, scale_layout), scale_layout == ATOM_SCALE_LAYOUT_REF, (uint8_t *)A_f6,
                          reinterpret_cast<hipStream_t>(stream));
}

size_t atom_f6_weight_bytes(int64_t N, int64_t K_total) {
  if (N < 1 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0) return 0;
  return (size_t)((K_total - kKeeper) / kGroup) * atom_f6_rows(N) * (104 + 4);
}

int atom_repack_weight_f6s(const void *B4, const void *sB, int64_t N, int64_t K_total, void *B_f6s, void *stream) {
  if (!B4 || !sB || !B_f6s) return ATOM_ERR_INVALID_ARG;
  if (N < 1 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0 || K_total > (1 << 20)) return ATOM_ERR_SHAPE;
  if (!aligned16(B4) || !aligned16(B_f6s)) return ATOM_ERR_ALIGN;
  const int G = (int)((K_total - kKeeper) / kGroup);
  RepackF6Pair pp;
  const int64_t Npad = (N + 255) / 256 * 256;
  pp.op[0] = RepackF6Params{(const uint8_t *)B4, (uint8_t *)B_f6s, N, Npad, (int)((K_total - kKeeper) / 2), G, nullptr, 0, 0,
                            (const half_t *)sB, reinterpret_cast<float *>((uint8_t *)B_f6s + (size_t)G * Npad * 104)};
  pp.op[1] = pp.op[0];
  hipLaunchKernelGGL(repack_f6_kernel, dim3((unsigned)(Npad / 256), (unsigned)G, 1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pp);
  return check_launch();
}

int atom_check_scale_pairs(const void *sB, int64_t G, int64_t N, int32_t *n_bad_dev, void *stream) {
  if (!sB || !n_bad_dev) return ATOM_ERR_INVALID_ARG;
  if (G < 1 || N < 2 || (N % 2) != 0 || G > (1 << 20) || N > (1 << 24)) return ATOM_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(sB) & 3u) || (reinterpret_cast<uintptr_t>(n_bad_dev) & 3u)) return ATOM_ERR_ALIGN;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(n_bad_dev, 0, sizeof(int32_t), s) != hipSuccess) return ATOM_ERR_LAUNCH;
  return launch_check_scale_pairs((const half_t *)sB, G, N, n_bad_dev, s);
}

int atom_verify_weight_f6s(const void *B4, const void *sB, int64_t N, int64_t K_total, const void *B_f6s, int32_t *n_bad_dev, void *stream) {
  if (!B4 || !sB || !B_f6s || !n_bad_dev) return ATOM_ERR_INVALID_ARG;
  if (N < 1 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0 || K_total > (1 << 20)) return ATOM_ERR_SHAPE;
  if (!aligned16(B4) || !aligned16(B_f6s) || (reinterpret_cast<uintptr_t>(n_bad_dev) & 3u)) return ATOM_ERR_ALIGN;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(n_bad_dev, 0, sizeof(int32_t), s) != hipSuccess) return ATOM_ERR_LAUNCH;
  return launch_verify_weight_f6s((const uint8_t *)B4, (const half_t *)sB, N, (int)((K_total - kKeeper) / 2), (int)((K_total - kKeeper) / kGroup),
                                  (const uint8_t *)B_f6s, n_bad_dev, s);
}

}  // extern "C"
