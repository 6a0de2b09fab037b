// INT4 paged KV cache for Atom on gfx950: append (prefill / decode) and batch-decode attention.
//
//   atom_kv_append_i4     <- FlashInferInitKvKernel_i4 / FlashInferAppendKvKernel_i4
//                            (e2e/punica-atom/punica/ops/csrc/flashinfer_adapter/flashinfer_impl.cuh:48-96 ->
//                             AppendPagedKVCachePrefillKernel / ...DecodeKernel, kernels/include/flashinfer/page.cuh:119-227)
//   atom_batch_decode_i4  <- FlashInferBatchDecodeKernel_i4 (flashinfer_impl.cuh:9-46 ->
//                             BatchDecodeWithPagedKVCacheKernel, kernels/include/flashinfer/decode.cuh:480-676)
//
// Layouts are the reference's (punica/utils/kvcache.py:17-26, page.cuh:78-110):
//   kv_data  u8  [pages, L, 2, N, P, D/2]   asymmetric u4, element 2j in the low nibble; value = nibble*scale - zero
//   kv_param f16 [pages, L, 2, N, P, 2]     (scale, zero) per token and head            (quantization.cuh:59-84)
//   sequence b owns pages kv_indices[kv_indptr[b] .. kv_indptr[b+1]); its length is
//   (npages-1)*P + last_page_offset[b].
// Both ops are pure HBM streaming (68 bytes per token, head and K/V); D = 128 as in the reference (CHECK_EQ(head_dim,128),
// punica_ops.cc:112), P a multiple of 16.
//
// Append: the reference walks a sequence's new tokens in a serial loop per (sequence, head) block; here every token is
// one workgroup (binary search of its sequence in append_indptr), 16-byte chunks per thread.
//
// Decode: one wave per (sequence, head, KV split); a 16-token tile of one page and head is exactly 1 KiB of K and 1 KiB
// of V, i.e. one coalesced wave access.  Lane (t, u) = (token of the tile, quarter): it loads K as two 8-byte pieces,
// dims [16u,16u+16) and [64+16u, 64+16u+16) -- the two halves of its 16 RoPE pairs -- and V as one 16-byte piece.
// RoPE is never applied to the keys: <R(pq) q, R(j) k> = <R(pq - j) q, k>, so each lane keeps q rotated by (pq - j) for
// ITS token j and turns it back by the constant angle 16*f_i per tile (4 FMAs per pair, no sincos in the loop).
// De-quantisation is folded: the zero point of V is accumulated as a scalar (o = sum p*s*u - sum p*z).  Every quad runs
// its own online softmax over its token subsequence; the 16 quads (and the KV splits, through a small FP32 workspace)
// are merged once at the end.
#include <math.h>

#include <cstdlib>
#include "common.h"

namespace atom {

constexpr int kHeadDim = 128;
constexpr float kLog2e = 1.4426950408889634f;

struct KvParams {
  uint8_t *data;
  half_t *param;
  const int32_t *indptr, *indices, *last_page_offset;
  int batch, L, layer, N, P;
};

// ------------------------------------------------------------------------------------------------ append
struct AppendParams {
  KvParams kv;
  const uint8_t *k, *v;
  const half_t *k_param, *v_param;
  const int32_t *append_indptr;   // NULL: one token per sequence
  int64_t T;
};

__global__ __launch_bounds__(256) void kv_append_kernel(AppendParams p) {
  const int64_t t = blockIdx.x;
  const int tid = threadIdx.x;
  int b, j, n;
  if (p.append_indptr) {
    int lo = 0, hi = p.kv.batch;                       // largest b with append_indptr[b] <= t
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (p.append_indptr[mid] <= t) lo = mid; else hi = mid;
    }
    b = lo;
    j = (int)(t - p.append_indptr[b]);
    n = p.append_indptr[b + 1] - p.append_indptr[b];
  } else {
    b = (int)t; j = 0; n = 1;
  }
  const int P = p.kv.P, N = p.kv.N;
  const int seq_len = (p.kv.indptr[b + 1] - p.kv.indptr[b] - 1) * P + p.kv.last_page_offset[b];
  const int pos = seq_len - n + j;                     // page.cuh:186-192
  if (pos < 0) return;
  const int64_t page = p.kv.indices[p.kv.indptr[b] + pos / P];
  const int e = pos % P;
  const int64_t base = (page * p.kv.L + p.kv.layer) * 2;             // [.., 2, N, P, ..]
  for (int i = tid; i < 2 * N * 4; i += 256) {                       // (k|v, head, 16-byte chunk)
    const int kv = i / (N * 4), h = (i >> 2) % N, ch = i & 3;
    const uint8_t *src = (kv ? p.v : p.k) + ((int64_t)t * N + h) * 64 + ch * 16;
    uint8_t *dst = p.kv.data + (((base + kv) * N + h) * P + e) * 64 + ch * 16;
    *reinterpret_cast<v4u *>(dst) = *reinterpret_cast<const v4u *>(src);
  }
  for (int i = tid; i < 2 * N; i += 256) {                           // (scale, zero) half2 per (k|v, head)
    const int kv = i / N, h = i % N;
    const unsigned *src = reinterpret_cast<const unsigned *>((kv ? p.v_param : p.k_param) + ((int64_t)t * N + h) * 2);
    unsigned *dst = reinterpret_cast<unsigned *>(p.kv.param + (((base + kv) * N + h) * P + e) * 2);
    *dst = *src;
  }
}

// Decode step, fused: the FP32 sums of the k / v projections (decode-batch GEMM, atom_gemm_w4a4_f32) are quantised per
// head (the u4 epilogue of the reference's _o4 GEMM: scale = (max-min)/15, zero = -min, q = clamp(round_half_away((x+zero)
// * (1/scale)), 0, 15), DenseLayerGEMM_i4_o4.cu:704-788) and written straight into the last token's slot of the paged
// cache (FlashInferAppendKvKernel_i4, flashinfer_impl.cuh:72-96) -- one launch instead of two epilogues + an append.
// One workgroup per sequence, half a wave per (K|V, head), 4 values per lane.
struct QuantAppendParams {
  KvParams kv;
  const float *k, *v;   // [batch, N * 128]
};

// Half a wave per (sequence, K / V, head) vector; the grid covers them all at once.  (Round 2 ran one workgroup per SEQUENCE that
// walked its 2 N head vectors eight at a time: 5.6 us at batch 1 -- a chain of 8 dependent round trips on one CU.)
// the u4 epilogue on one 128-value head vector held by 32 lanes (4 values each; both halves of a wave at once): returns the lane's four
// codes as 16 bits and the vector's (scale, zero) as a half2 bit pattern -- what the cache stores
__device__ __forceinline__ unsigned short quant_head_u4(const v4f &x, unsigned &sz) {
  float lo = fminf(fminf(x[0], x[1]), fminf(x[2], x[3])), hi = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
#pragma unroll
  for (int k = 16; k >= 1; k >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, k));
    hi = fmaxf(hi, __shfl_xor(hi, k));
  }
  const float scale = (hi - lo) / 15.f, zero = -lo, rs = 1.0f / scale;
  unsigned w = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t = (x[k] + zero) * rs;
    float tr = truncf(t);
    if (fabsf(t - tr) >= 0.5f) tr += copysignf(1.0f, t);
    tr = fminf(fmaxf(tr, 0.f), 15.f);
    if (scale == 0.f) tr = 0.f;
    w |= (unsigned)(int)tr << (4 * k);
  }
  sz = (unsigned)__builtin_bit_cast(unsigned short, f2h(scale)) | ((unsigned)__builtin_bit_cast(unsigned short, f2h(zero)) << 16);
  return (unsigned short)w;
}

__global__ __launch_bounds__(256) void kv_quant_append_kernel(QuantAppendParams p) {
  const int l = threadIdx.x & 31;
  const int P = p.kv.P, N = p.kv.N;
  const int64_t vec = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);   // (b, kv, h)
  if (vec >= (int64_t)p.kv.batch * 2 * N) return;
  const int b = (int)(vec / (2 * N)), gi = (int)(vec % (2 * N));
  const int kv = gi / N, h = gi % N;
  const v4f x = *reinterpret_cast<const v4f *>((kv ? p.v : p.k) + ((int64_t)b * N + h) * kHeadDim + 4 * l);   // (independent of the page walk)
  const int seq_len = (p.kv.indptr[b + 1] - p.kv.indptr[b] - 1) * P + p.kv.last_page_offset[b];
  const int pos = seq_len - 1;
  if (pos < 0) return;
  const int64_t page = p.kv.indices[p.kv.indptr[b] + pos / P];
  const int e = pos % P;
  const int64_t base = (page * p.kv.L + p.kv.layer) * 2;             // [.., 2, N, P, ..]
  unsigned sz;
  const unsigned short w = quant_head_u4(x, sz);
  const int64_t slot = ((base + kv) * N + h) * P + e;
  *reinterpret_cast<unsigned short *>(p.kv.data + slot * 64 + 2 * l) = w;
  if (l == 0) *reinterpret_cast<unsigned *>(p.kv.param + slot * 2) = sz;
}

// ------------------------------------------------------------------------------------------------ decode
// x / d for 0 <= x < 2^31 by a multiplication with a reciprocal the HOST prepared: s = ceil(log2 d), m = ceil(2^(31+s) / d) < 2^32,
// q = mulhi(x, m) >> (s - 1) -- exact, since x * (m d - 2^(31+s)) < 2^31 * 2^s; d = 1: m = 0, q = x.  (The decode kernel's prologue held
// seven 32-bit divisions by run-time values -- heads, splits, tiles per page, page length --, ~30 instructions and a VALU -> SALU round
// trip each, two of them between the dependent loads of the page-table chain.)
struct FastDiv {
  unsigned m, sh;
};
static FastDiv make_fastdiv(unsigned d) {
  if (d <= 1) return FastDiv{0u, 0u};
  unsigned s = 0;
  while ((1u << s) < d) ++s;
  return FastDiv{(unsigned)((((unsigned long long)1 << (31 + s)) + d - 1) / d), s - 1};
}
__device__ __forceinline__ int fdiv(int x, FastDiv f) { return f.m ? (int)(__umulhi((unsigned)x, f.m) >> f.sh) : x; }

struct DecodeParams {
  KvParams kv;
  FastDiv dN, dsplits, dtpp, dP;   // reciprocals of kv.N, splits, kv.P / 16, kv.P
  int ppw;                         // WGM: (sequence, head) pairs per workgroup = kWgmWaves / splits
  const half_t *q;      // [B, N, 128]
  half_t *o;            // [B, N, 128]
  float *ws;            // splits > 1: [B, N, splits, 130] = o[128] (not normalised), m, d
  int splits;
  float sm_scale, log2_theta, rope_inv_scale;
  const float *k32, *v32;   // optional (atom_batch_decode_append_i4): the step's k / v projections as FP32 sums [B, N * 128] -- quantised and
};                          // appended as the last token by the wave that attends to it

// 8 packed u4 (nibble e at bits 4e) -> 4 half2 registers holding 1024 + nibble: m[k] = {1024 + n_k, 1024 + n_(k+4)}.
// (x >> 4k) & 0x000F000F | 0x64006400 is one v_and_or_b32 (plus one shift for k > 0): 7 instructions per 8 elements,
// and v_fma_mix_f32 reads the halves directly.  The bias 1024 is removed once per token, never per element.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nib8h(unsigned x, h2 (&m)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned r;                                        // hipcc emits v_and + v_or for the C expression
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x >> (4 * k)), "s"(0x000F000Fu), "v"(0x64006400u));
    m[k] = __builtin_bit_cast(h2, r);
  }
}
constexpr float kNibBias = 1024.0f;

__device__ __forceinline__ float quad_sum_f(float x) {
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  return x;
}

// sin / cos of 2*pi*rev (v_sin_f32 / v_cos_f32 take revolutions; the reference uses the __sincosf fast intrinsic too,
// decode.cuh:63-66)
__device__ __forceinline__ void sincos_rev(float rev, float &s, float &c) {
  const float fr = rev - floorf(rev);
  s = __builtin_amdgcn_sinf(fr);
  c = __builtin_amdgcn_cosf(fr);
}

struct TileRegs {     // one 16-token tile's operands of this lane, as loaded
  v2u k1, k2;         // K dims [16u,16u+16) and [64+16u, 64+16u+16)
  v4u v;              // V dims [32u, 32u+32)
  unsigned kq, vq;    // (scale, zero) half2 of my token
};

// Everything about a tile's addresses except the page number is loop-invariant: per wave (layer, head) and per lane
// (token slot, quarter).  Per tile that leaves one 64-bit scalar multiply-add per region instead of the ~130 SALU + 36
// VALU of the straightforward indexing.
struct TileBase {
  const uint8_t *k;          // data  + (layer, K, head) offset;  V is + vdelta
  const half_t *kq;          // param + (layer, K, head) offset;  V is + vqdelta
  int64_t page_bytes, page_halves, vdelta, vqdelta;
  int lk, lv, lq;            // per-lane byte / half offsets inside a 16-token tile
};

__device__ __forceinline__ TileRegs load_tile(const TileBase &tb, int64_t page, int sub) {
  const uint8_t *kp = tb.k + page * tb.page_bytes + sub * (16 * 64);
  const half_t *qp = tb.kq + page * tb.page_halves + sub * (16 * 2);
  TileRegs r;
  r.k1 = *reinterpret_cast<const v2u *>(kp + tb.lk);
  r.k2 = *reinterpret_cast<const v2u *>(kp + tb.lk + 32);
  r.v = *reinterpret_cast<const v4u *>(kp + tb.vdelta + tb.lv);
  r.kq = *reinterpret_cast<const unsigned *>(qp + tb.lq);
  r.vq = *reinterpret_cast<const unsigned *>(qp + tb.vqdelta + tb.lq);
  return r;
}

// WGM (round 6): the KV splits of a (sequence, head) pair are WAVES of one workgroup and their partial states meet in LDS -- no workspace,
// no merge launch (4.7 us of a decode step at batch 16).  A workgroup is always 12 waves = a CU's resident set at this kernel's
// registers: 12 / splits pairs x splits (a 6-wave workgroup per pair left every second wave slot empty -- the second one does not fit
// beside the first's 2 + 2 + 1 + 1 waves per SIMD -- and cost 38.5 us instead of 27.3: profiles/r06/ab_decode_ring.txt).  Used when
// pairs alone fill the chip and splits divides 12 (batch_decode_impl).  The merge below is decode_merge_kernel's, operation for
// operation: same bits.
constexpr int kWgmWaves = 12;
#ifndef ATOM_DECODE_DP
#define ATOM_DECODE_DP 2
#endif
// INNER (round 6, !WGM): the splits come in workgroups of INNER waves whose partial states meet in LDS and leave as ONE partial state
// (un-normalised values, running maximum, denominator: the state a single wave writes) -- a batch-1 step at context 1024 then hands 4
// partial states per head to the merge (or to o_proj's launch: atom_gemm_w4a4_multi_merge_q) instead of 16.  p.splits counts waves.
#ifndef ATOM_DECODE_KINNER
#define ATOM_DECODE_KINNER 4
#endif
constexpr int kInner = ATOM_DECODE_KINNER;
template <bool WGM, int INNER = 1>
__global__ __launch_bounds__(WGM ? 64 * kWgmWaves : 64 * INNER, 3) void batch_decode_kernel(DecodeParams p) {
  const int lane = threadIdx.x & 63;
  const int t = lane >> 2, u = lane & 3;
  {  // the kernel arguments the prologue dereferences, in ONE batch of scalar loads: the page-table chain below (indptr -> indices ->
     // tiles) is three dependent trips as it is; hipcc fetched last_page_offset's pointer a trip late
    const void *a0 = p.kv.indptr, *a1 = p.kv.indices, *a2 = p.kv.last_page_offset, *a3 = p.q, *a4 = p.kv.data, *a5 = p.kv.param, *a6 = p.k32, *a7 = p.v32;
    asm volatile("" ::"s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(a6), "s"(a7));
  }
  const int N = p.kv.N, P = p.kv.P;
  int pair = blockIdx.x, sp = blockIdx.y, lp = 0;       // (sequence, head) pair, KV split, pair within the workgroup
  if constexpr (!WGM && INNER > 1) sp = (int)blockIdx.y * INNER + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  bool live = true;
  if constexpr (WGM) {
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    lp = fdiv(w, p.dsplits);
    sp = w - lp * p.splits;
    pair = (int)blockIdx.x * p.ppw + lp;
    live = pair < p.kv.batch * N;                       // (the last workgroup's spare waves: no tiles, no writes, the barriers only)
    pair = min(pair, p.kv.batch * N - 1);
  }
  const int b = fdiv(pair, p.dN), h = pair - b * N;
  const int pg0 = p.kv.indptr[b];
  const int seq_len = live ? (p.kv.indptr[b + 1] - pg0 - 1) * P + p.kv.last_page_offset[b] : 0;
  const int ntiles = (max(seq_len, 0) + 15) >> 4;       // (an empty sequence: no tiles, no page-table read, output zeros)
  const int chunk = fdiv(ntiles + p.splits - 1, p.dsplits);
  const int tile0 = sp * chunk, tile1 = min(ntiles, tile0 + chunk);
  const int tpp = P >> 4;                               // 16-token tiles per page

  // two tiles in flight ahead of the one being computed, and the page index (a dependent scalar load) one tile ahead of
  // that: otherwise every iteration waits for the page table before it can issue its vector loads
  TileBase tb;
  {
    const int64_t blk = (int64_t)N * P;                  // tokens x heads of one (page, layer, K|V) block
    tb.k = p.kv.data + ((int64_t)p.kv.layer * 2 * N + h) * P * 64;
    tb.kq = p.kv.param + ((int64_t)p.kv.layer * 2 * N + h) * P * 2;
    tb.page_bytes = (int64_t)p.kv.L * 2 * blk * 64;
    tb.page_halves = (int64_t)p.kv.L * 2 * blk * 2;
    tb.vdelta = blk * 64;
    tb.vqdelta = blk * 2;
    tb.lk = t * 64 + 8 * u;
    tb.lv = t * 64 + 16 * u;
    tb.lq = t * 2;
  }
  // Fused append (round 6; p.k32 != NULL): this step's k / v of head h arrive as FP32 sums.  The ONE wave whose tile range holds the
  // last token quantises them (lanes 0-31 K, 32-63 V: kv_quant_append_kernel's arithmetic), writes the cache slot, and hands the 64 + 64
  // code bytes and the two (scale, zero) pairs to the lane of that token through LDS -- the attention below never waits for the store
  // to come back from memory, and the step saves a launch.  Same cache bytes, same output as append -> decode.
  constexpr int LP = WGM ? kWgmWaves / 2 : 1;           // pairs per workgroup at most
  __shared__ unsigned short stash_q[LP][2][32];
  __shared__ unsigned stash_sz[LP][2];
  const int last_tile = (seq_len - 1) >> 4;
  const bool appends = p.k32 != nullptr && seq_len > 0 && last_tile >= tile0 && last_tile < tile1;   // wave-uniform
  if (appends) {
    const int kvh = lane >> 5, l = lane & 31;
    const v4f x = *reinterpret_cast<const v4f *>((kvh ? p.v32 : p.k32) + ((int64_t)b * N + h) * kHeadDim + 4 * l);
    unsigned sz;
    const unsigned short w = quant_head_u4(x, sz);
    const int pos = seq_len - 1;
    const int ppage = fdiv(pos, p.dP);
    const int64_t slot = ((((int64_t)p.kv.indices[pg0 + ppage] * p.kv.L + p.kv.layer) * 2 + kvh) * N + h) * P + (pos - ppage * P);
    *reinterpret_cast<unsigned short *>(p.kv.data + slot * 64 + 2 * l) = w;
    if (l == 0) *reinterpret_cast<unsigned *>(p.kv.param + slot * 2) = sz;
    stash_q[lp][kvh][l] = w;
    if (l == 0) stash_sz[lp][kvh] = sz;
  }
  __syncthreads();
  // ---- the tile pipeline (round 6, second form).  The ISA of rounds 1-5 defeated its own prefetch: the page-table entry of the tile three
  // ahead was a VECTOR load (the kernel also writes the cache, so hipcc will not use the scalar cache for kv_indices) consumed through
  // v_readfirstlane behind s_waitcnt vmcnt(0) at the top of every iteration, and `cur = nxt` copied registers whose loads were still in
  // flight (another vmcnt(0)): every iteration waited for everything it had requested.  Now: (a) the page entries of this wave's whole
  // tile range arrive in ONE vector load -- lane i holds the entry of page p0 + i, read per tile with v_readlane (a second load only
  // past 64 pages) --; (b) DP tile buffers used in rotation, no copies (the loop is unrolled by DP; 2 measured best, 3-5 within 1-4 %:
  // profiles/r06/ab_decode_ring2.txt); (c) the refill of a buffer is
  // UNCONDITIONAL -- past the wave's last tile it re-reads that tile (resident lines) -- so that every trip issues the same requests
  // and hipcc's waits are exact counts (csrc/gemvq_w4a4.hip learned the same: requests under run-time conditions end as vmcnt(0)).
  const int nt = max(tile1 - tile0, 0);
  const int p0 = fdiv(tile0, p.dtpp);                   // first page of my range (relative to pg0)
  const int plast = nt > 0 ? fdiv(tile1 - 1, p.dtpp) : p0;
  int pseg = 0;                                          // 64-page segment held in pgv
  int pgv = 0;
  if (nt > 0) pgv = p.kv.indices[pg0 + min(p0 + lane, plast)];
  // the refill cursor: tile rt = (page rp relative to p0, sub-tile rs), advancing until my last tile
  int rt = tile0, rp = 0, rs = tile0 - p0 * tpp;
  auto advance = [&]() {
    if (rt + 1 < tile1) {
      ++rt;
      if (++rs == tpp) { rs = 0; ++rp; }
    }
  };
  auto page_at = [&](int prel) -> int64_t {               // (uniform) entry of page p0 + prel
    if ((prel >> 6) != pseg) {                          // beyond the 64 entries held: the next segment (once per 1024+ tokens of a wave)
      pseg = prel >> 6;
      pgv = p.kv.indices[pg0 + min(p0 + pseg * 64 + lane, plast)];
      asm volatile("" : "+v"(pgv));                      // (waited for HERE, inside the branch: pending at the join it would cost every tile a vmcnt(0))
    }
    return (int64_t)__builtin_amdgcn_readlane(pgv, prel & 63);
  };
  constexpr int DP = ATOM_DECODE_DP;                     // tile buffers (tools builds: -DATOM_DECODE_DP=n)
  TileRegs ring[DP];
  if (nt > 0) {
#pragma unroll
    for (int u = 0; u < DP; ++u) {
      ring[u] = load_tile(tb, page_at(rp), rs);
      advance();
    }
  }

  // q rotated to the relative position of MY token of the first tile: A = R((len-1 - j) f) q, pairs (i, i+64)
  // (round 6: kept as register PAIRS -- A[i] = {A1, A2}, SC[i] = {sin, cos} of the 16-position step -- so that the per-tile rotation is
  // two packed FP32 instructions per pair instead of four scalar ones, and the sum of A sixteen packed adds instead of 32)
  v2f A[16], SC[16];
  // The frequency of pair j and the sin / cos of its 16-position step depend on j alone: lane j computes them for j = lane (one exp2, one
  // sin, one cos instead of 16 + 32 per lane -- the 16 token lanes of a quarter used to repeat each other's work, ~45 transcendental and
  // ~150 other instructions of a prologue that was a fifth of the wave's instructions at 11 tiles) and the wave shares them through its own
  // 768 bytes of LDS: no workgroup barrier, a wave's LDS operations complete in order.  Same expressions on the same inputs: same bits.
  constexpr int RW = WGM ? kWgmWaves : INNER;
  __shared__ float rope_fr[RW][64];
  __shared__ v2f rope_sc[RW][64];
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  {
    // decode.cuh:535-539: freq = rope_inv_scale * theta^(-2 (i mod 64) / 128); here in revolutions
    const float fr = p.rope_inv_scale * 0.15915494309189535f *
                     __builtin_amdgcn_exp2f(-p.log2_theta * (float)(2 * lane) * (1.0f / kHeadDim));
    float s16, c16;
    sincos_rev(16.0f * fr, s16, c16);
    rope_fr[wv][lane] = fr;
    rope_sc[wv][lane] = v2f{s16, c16};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  {
    const half_t *qp = p.q + ((int64_t)b * N + h) * kHeadDim;
    v4u r1[2], r2[2];
    r1[0] = *reinterpret_cast<const v4u *>(qp + 16 * u);
    r1[1] = *reinterpret_cast<const v4u *>(qp + 16 * u + 8);
    r2[0] = *reinterpret_cast<const v4u *>(qp + 64 + 16 * u);
    r2[1] = *reinterpret_cast<const v4u *>(qp + 64 + 16 * u + 8);
    const half_t *h1 = reinterpret_cast<const half_t *>(r1), *h2p = reinterpret_cast<const half_t *>(r2);
    const float delta = (float)((seq_len - 1) - (tile0 * 16 + t));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float fr = rope_fr[wv][16 * u + i];
      float s, c;
      sincos_rev(delta * fr, s, c);
      const float q1 = (float)h1[i], q2 = (float)h2p[i];
      A[i] = v2f{q1 * c - q2 * s, q2 * c + q1 * s};
      SC[i] = rope_sc[wv][16 * u + i];
    }
  }

  float o[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) o[i] = 0.f;
  float m = -INFINITY, d = 0.f, zacc = 0.f;
  const float qk_scale = p.sm_scale * kLog2e;           // decode.cuh:500: softmax in base 2

  for (int base = tile0; base < tile1; base += DP) {
#pragma unroll
  for (int ru = 0; ru < DP; ++ru) {
    const int tile = base + ru;
    TileRegs &r = ring[ru];
    if (tile < tile1) {                                  // (wave-uniform; the last trip may be partial)
    const bool valid = tile * 16 + t < seq_len;
    if (appends && tile == last_tile && t == ((seq_len - 1) & 15)) {   // my token is the one appended above: its bytes come from LDS
      const char *sk = reinterpret_cast<const char *>(&stash_q[lp][0][0]), *sv = reinterpret_cast<const char *>(&stash_q[lp][1][0]);
      r.k1 = *reinterpret_cast<const v2u *>(sk + 8 * u);
      r.k2 = *reinterpret_cast<const v2u *>(sk + 32 + 8 * u);
      r.v = *reinterpret_cast<const v4u *>(sv + 16 * u);
      r.kq = stash_sz[lp][0];
      r.vq = stash_sz[lp][1];
    }

    // score = sum over my 16 pairs of (u1*ks - kz) * A1 + (u2*ks - kz) * A2 = ks * sum((1024+u) . A) - (kz + 1024 ks) * sum(A)
    float acc = 0.f;
    v2f sum2 = v2f{0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      h2 m1[4], m2[4];
      nib8h(r.k1[w], m1);
      nib8h(r.k2[w], m2);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc = __builtin_fmaf((float)m1[k].x, A[8 * w + k].x, acc);
        acc = __builtin_fmaf((float)m1[k].y, A[8 * w + 4 + k].x, acc);
        acc = __builtin_fmaf((float)m2[k].x, A[8 * w + k].y, acc);
        acc = __builtin_fmaf((float)m2[k].y, A[8 * w + 4 + k].y, acc);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) sum2 += A[i];             // v_pk_add_f32
    const float sumA = sum2.x + sum2.y;
    const float ks = (float)__builtin_bit_cast(half_t, (unsigned short)(r.kq & 0xFFFF));
    const float kz = (float)__builtin_bit_cast(half_t, (unsigned short)(r.kq >> 16));
    float s = quad_sum_f(__builtin_fmaf(acc, ks, -(__builtin_fmaf(kNibBias, ks, kz) * sumA))) * qk_scale;
    if (!valid) s = -INFINITY;
    // online softmax of this quad's token subsequence (decode.cuh update_partial_state / state.cuh)
    const float mn = fmaxf(m, s);
    if (__any(mn > m)) {                               // rescale only when some running maximum moved
      const float alpha = mn == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f(m - mn);
      d *= alpha;
      zacc *= alpha;
#pragma unroll
      for (int i = 0; i < 32; ++i) o[i] *= alpha;
      m = mn;
    }
    const float pr = valid ? __builtin_amdgcn_exp2f(s - m) : 0.f;
    d += pr;
    // entries past the sequence end are uninitialised memory (possibly NaN / Inf parameters): they must contribute
    // exactly nothing (the reference guards the whole update, decode.cuh:153)
    const unsigned vq = valid ? r.vq : 0u;
    const float vs = (float)__builtin_bit_cast(half_t, (unsigned short)(vq & 0xFFFF));
    const float vz = (float)__builtin_bit_cast(half_t, (unsigned short)(vq >> 16));
    const float ps = pr * vs;
    zacc = __builtin_fmaf(pr, __builtin_fmaf(kNibBias, vs, vz), zacc);     // o = sum ps*(1024+u) - sum pr*(vz + 1024 vs)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      h2 mv[4];
      nib8h(r.v[w], mv);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[8 * w + k] = __builtin_fmaf(ps, (float)mv[k].x, o[8 * w + k]);
        o[8 * w + 4 + k] = __builtin_fmaf(ps, (float)mv[k].y, o[8 * w + 4 + k]);
        // one v_fma_mix_f32 each: without the anchors hipcc converts the halves first and SLP-packs (cvt + pk_fma + moves)
        asm volatile("" : "+v"(o[8 * w + k]), "+v"(o[8 * w + 4 + k]));
      }
    }
    // my token of the next tile is 16 positions later: A <- R(-16 f) A = (A1 C + A2 S, A2 C - A1 S)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      // {fma(a1, C, a2 * S), fma(a1, -S, a2 * C)}: the scalar form's four operations, bit for bit (fma(-a1, S, x) = fma(a1, -S, x))
      v2f tt;
      asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=&v"(tt) : "v"(A[i]), "v"(SC[i]));
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "=v"(A[i]) : "v"(A[i]), "v"(SC[i]), "v"(tt));
    }
    }
    r = load_tile(tb, page_at(rp), rs);                  // this buffer's next tenant: DP tiles on (my last tile again at the end)
    advance();
  }
  }

  // merge the 16 quads: lanes with equal u hold the same 32 output dims.  Butterfly over lane bits 5, 4, 3, 2 without the LDS
  // pipeline (35 values x 4 stages were 140 ds_bpermute per wave): v_permlane32_swap / v_permlane16_swap bring lane i ^ 32
  // (i ^ 16) alongside; after those the values repeat with period 16 (then 8) over the lanes, so lane i ^ 8 (i ^ 4) holds what
  // lane (i + 8) (i + 4) of the same 16-lane row holds: DPP row rotations.
  auto over_quads = [](float x, auto op) {
    {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
      x = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
      x = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    x = op(x, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x128, 0xF, 0xF, true)));   // row_ror:8
    x = op(x, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x124, 0xF, 0xF, true)));   // row_ror:4
    return x;
  };
  auto fadd = [](float a, float b) { return a + b; };
  auto fmax_ = [](float a, float b) { return fmaxf(a, b); };
  const float mall = over_quads(m, fmax_);
  const float sc = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mall);
  d = over_quads(d * sc, fadd);
  zacc = over_quads(zacc * sc, fadd);
  // The 32 values per lane: the same butterfly as a REDUCE-SCATTER (round 6).  Stage 1 (lane ^ 32) swaps a value of the half I give away
  // against my partner's value of the half I keep -- v_permlane32_swap(o[i], o[16 + i]) leaves {mine, partner's} of dims i in lanes
  // 0-31 and of dims 16 + i in lanes 32-63 --, stage 2 (lane ^ 16) halves again, stages 3, 4 run on the 8 values left: 16 + 8 swaps and
  // 16 + 8 + 16 additions instead of 64 + 128.  Every sum pairs the same two operands in the same stage order as over_quads: same bits.
  // Afterwards lane (t, u) holds og[i] = dim 32 u + 8 (t >> 2) + i -- in every lane of its (t >> 2) group.
  float og[8];
  {
    float y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[i] * sc), __float_as_uint(o[16 + i] * sc), false, false);
      y[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(y[i]), __float_as_uint(y[8 + i]), false, false);
      float z = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      z = z + __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(z), 0x128, 0xF, 0xF, true));   // row_ror:8
      z = z + __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(z), 0x124, 0xF, 0xF, true));   // row_ror:4
      og[i] = z - zacc;                                      // sum p*(s*u - z)
    }
  }
  const int tg = t >> 2;                                     // my dims: 32 u + 8 tg .. + 7
  if constexpr (WGM) {
    // partial states [wave][130] through LDS, then decode_merge_kernel's arithmetic: 128 threads per pair
    __shared__ float part[kWgmWaves][kHeadDim + 2];
    const int w = lp * p.splits + sp;
    if ((t & 3) == 0) {
      *reinterpret_cast<v4f *>(&part[w][32 * u + 8 * tg]) = v4f{og[0], og[1], og[2], og[3]};
      *reinterpret_cast<v4f *>(&part[w][32 * u + 8 * tg + 4]) = v4f{og[4], og[5], og[6], og[7]};
      if (lane == 0) {
        part[w][kHeadDim] = mall;
        part[w][kHeadDim + 1] = d;
      }
    }
    __syncthreads();
    const int mp = (int)threadIdx.x >> 7, dim = (int)threadIdx.x & 127;
    const int mpair = (int)blockIdx.x * p.ppw + mp;
    if (mp >= p.ppw || mpair >= p.kv.batch * N) return;
    const float(*pp)[kHeadDim + 2] = &part[mp * p.splits];
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < kWgmWaves; ++s)
      if (s < p.splits) M = fmaxf(M, pp[s][kHeadDim]);
    float acc = 0.f, den = 0.f;
#pragma unroll
    for (int s = 0; s < kWgmWaves; ++s)
      if (s < p.splits) {
        const float mv = pp[s][kHeadDim];
        const float wgt = mv == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mv - M);
        acc = __builtin_fmaf(pp[s][dim], wgt, acc);
        den = __builtin_fmaf(pp[s][kHeadDim + 1], wgt, den);
      }
    p.o[(int64_t)mpair * kHeadDim + dim] = (half_t)(den > 0.f ? acc / den : 0.f);
    return;
  }
  if constexpr (INNER > 1) {
    // the INNER waves' states through LDS; 128 threads fold them into one partial state in wave order: M = max m_s, w_s = 2^(m_s - M),
    // O = sum o_s w_s, D = sum d_s w_s (the merge's own operations, without its final division)
    __shared__ float ipart[INNER][kHeadDim + 2];
    const int w = (int)threadIdx.x >> 6;
    if ((t & 3) == 0) {
      *reinterpret_cast<v4f *>(&ipart[w][32 * u + 8 * tg]) = v4f{og[0], og[1], og[2], og[3]};
      *reinterpret_cast<v4f *>(&ipart[w][32 * u + 8 * tg + 4]) = v4f{og[4], og[5], og[6], og[7]};
      if (lane == 0) {
        ipart[w][kHeadDim] = mall;
        ipart[w][kHeadDim + 1] = d;
      }
    }
    __syncthreads();
    const int dim = threadIdx.x;
    if (dim >= kHeadDim) return;
    float M = -INFINITY;
#pragma unroll
    for (int s2 = 0; s2 < INNER; ++s2) M = fmaxf(M, ipart[s2][kHeadDim]);
    float acc = 0.f, den = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < INNER; ++s2) {
      const float mv = ipart[s2][kHeadDim];
      const float wgt = mv == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mv - M);
      acc = __builtin_fmaf(ipart[s2][dim], wgt, acc);
      den = __builtin_fmaf(ipart[s2][kHeadDim + 1], wgt, den);
    }
    const int outer = p.splits / INNER;
    if (outer == 1) {                                    // the only partial state of its pair: the output
      p.o[((int64_t)b * N + h) * kHeadDim + dim] = (half_t)(den > 0.f ? acc / den : 0.f);
    } else {
      float *wp = p.ws + (((int64_t)b * N + h) * outer + blockIdx.y) * (kHeadDim + 2);
      wp[dim] = acc;
      if (dim == 0) {
        wp[kHeadDim] = M;
        wp[kHeadDim + 1] = den;
      }
    }
    return;
  }
  if ((t & 3) != 0) return;
  if (p.splits == 1) {
    const float rd = d > 0.f ? 1.0f / d : 0.f;
    v4u pk;
    half_t *hv = reinterpret_cast<half_t *>(&pk);
#pragma unroll
    for (int i = 0; i < 8; ++i) hv[i] = (half_t)(og[i] * rd);
    *reinterpret_cast<v4u *>(p.o + ((int64_t)b * N + h) * kHeadDim + 32 * u + 8 * tg) = pk;
  } else {
    float *wp = p.ws + (((int64_t)b * N + h) * p.splits + sp) * (kHeadDim + 2);
    *reinterpret_cast<v4f *>(wp + 32 * u + 8 * tg) = v4f{og[0], og[1], og[2], og[3]};
    *reinterpret_cast<v4f *>(wp + 32 * u + 8 * tg + 4) = v4f{og[4], og[5], og[6], og[7]};
    if (lane == 0) {
      wp[kHeadDim] = mall;
      wp[kHeadDim + 1] = d;
    }
  }
}

// out[b,h,:] = sum_s o_s * 2^(m_s - M) / sum_s d_s * 2^(m_s - M)
// Round 6: every split's (value, m, d) is requested before the first one is used -- in batches of 8 splits with compile-time bounds.
// The partial states were written by waves on other XCDs, so each request is a trip to memory (~1.5 us): the two run-time loops of
// rounds 1-5 (one for the maximum, one for the sums, their loads inside) paid it 2 x splits times in sequence on a 130-thread kernel
// (4.7 us per decode step for 133 KB).  Same operations in the same order: same bits.
// SB = splits handled in one batch of loads: 8 or 16 by the launcher.  (Round 6 lowered the KV-split size to 4 tiles: a batch-1 step at
// context 1024 has 16 splits, and with SB = 8 that took the run-time loops below -- four dependent trips to memory instead of one.)
template <int SB>
__global__ __launch_bounds__(128) void decode_merge_kernel(const float *ws, half_t *o, int splits) {
  const int64_t bh = blockIdx.x;
  const int dim = threadIdx.x;
  const float *wp = ws + bh * splits * (kHeadDim + 2);
  float M = -INFINITY;
  if (splits <= SB) {
    float ov[SB], mv[SB], dv[SB];
#pragma unroll
    for (int s = 0; s < SB; ++s) {
      const int sc = min(s, splits - 1);
      ov[s] = wp[sc * (kHeadDim + 2) + dim];
      mv[s] = wp[sc * (kHeadDim + 2) + kHeadDim];
      dv[s] = wp[sc * (kHeadDim + 2) + kHeadDim + 1];
    }
#pragma unroll
    for (int s = 0; s < SB; ++s)
      if (s < splits) M = fmaxf(M, mv[s]);
    float acc = 0.f, den = 0.f;
#pragma unroll
    for (int s = 0; s < SB; ++s)
      if (s < splits) {
        const float w = mv[s] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mv[s] - M);
        acc = __builtin_fmaf(ov[s], w, acc);
        den = __builtin_fmaf(dv[s], w, den);
      }
    o[bh * kHeadDim + dim] = (half_t)(den > 0.f ? acc / den : 0.f);
    return;
  }
  for (int s0 = 0; s0 < splits; s0 += SB) {
    float mv[SB];
#pragma unroll
    for (int s = 0; s < SB; ++s) mv[s] = wp[min(s0 + s, splits - 1) * (kHeadDim + 2) + kHeadDim];
#pragma unroll
    for (int s = 0; s < SB; ++s) M = fmaxf(M, mv[s]);        // (a clamped repeat of the last split changes no maximum)
  }
  float acc = 0.f, den = 0.f;
  for (int s0 = 0; s0 < splits; s0 += SB) {
    float ov[SB], mv[SB], dv[SB];
#pragma unroll
    for (int s = 0; s < SB; ++s) {
      const int sc = min(s0 + s, splits - 1);
      ov[s] = wp[sc * (kHeadDim + 2) + dim];
      mv[s] = wp[sc * (kHeadDim + 2) + kHeadDim];
      dv[s] = wp[sc * (kHeadDim + 2) + kHeadDim + 1];
    }
#pragma unroll
    for (int s = 0; s < SB; ++s)
      if (s0 + s < splits) {
        const float w = mv[s] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mv[s] - M);
        acc = __builtin_fmaf(ov[s], w, acc);
        den = __builtin_fmaf(dv[s], w, den);
      }
  }
  o[bh * kHeadDim + dim] = (half_t)(den > 0.f ? acc / den : 0.f);
}

static int check_kv(const void *kv_data, const void *kv_param, const int32_t *indptr, const int32_t *indices,
                    const int32_t *lpo, int batch, int L, int layer, int N, int P, int D) {
  if (!kv_data || !kv_param || !indptr || !indices || !lpo) return ATOM_ERR_INVALID_ARG;
  if (D != kHeadDim || batch < 1 || L < 1 || layer < 0 || layer >= L || N < 1 || P < 16 || (P % 16) != 0) return ATOM_ERR_SHAPE;
  if (!aligned16(kv_data) || (reinterpret_cast<uintptr_t>(kv_param) & 3u)) return ATOM_ERR_ALIGN;
  return ATOM_OK;
}

// KV splits: batch*heads*splits waves run in rounds of 3072 (12 resident waves per CU); a wave costs its tiles plus about
// two tiles of prologue (32 sincos) and merge, the merge kernel grows with the splits.  Pick the split count that minimises
// rounds x (tiles per wave + 2) under at least 8 tiles per split (measured: profiles/r01_kv_decode.txt).
static int decode_splits_total(int batch, int N, int max_pages, int P) {
  if (max_pages <= 0) return 1;
  if (const int forced = ATOM_TUNE("ATOM_DECODE_SPLITS", 0)) return forced;
  const int64_t tiles = (int64_t)max_pages * (P / 16), pairs = (int64_t)batch * N;
  const int min_tiles = ATOM_TUNE("ATOM_DECODE_MIN_TILES", 4);   // (round 6: 4 -- a wave alone on its SIMD issues one VALU per ~7 cycles, so at small batches twice the waves with half the tiles each win: 11.0 -> 9.0 us at batch 1, context 1024)
  int64_t smax = tiles / min_tiles;
  if (smax > 64) smax = 64;
  int best = 1;
  double best_cost = 1e30;
  for (int64_t s = 1; s <= (smax < 1 ? 1 : smax); ++s) {
    const int64_t rounds = (pairs * s + 3071) / 3072;
    const double cost = (double)rounds * ((double)tiles / (double)s + 2.0) + 0.15 * (double)s;
    if (cost < best_cost - 1e-9) { best_cost = cost; best = (int)s; }
  }
  return best;
}

// The plan of a decode call: waves per (sequence, head) pair, and how many of them share a workgroup and leave ONE partial state
// (kInner where the pairs do not fill the chip and the count divides; the pairs-fill-the-chip case merges ALL of a pair's waves in
// one workgroup: batch_decode_impl's wgm).  decode_splits() = partial states per pair as the merge / the workspace see them.
static int decode_inner(int batch, int N, int total) {
  if ((int64_t)batch * N >= ATOM_TUNE("ATOM_DECODE_WGM_PAIRS", 256) && kWgmWaves % total == 0) return 1;   // (wgm)
  return (ATOM_TUNE("ATOM_DECODE_INNER", 1) && total >= kInner && total % kInner == 0) ? kInner : 1;
}
static int decode_splits(int batch, int N, int max_pages, int P) {
  const int total = decode_splits_total(batch, N, max_pages, P);
  return total / decode_inner(batch, N, total);
}

}  // namespace atom

using namespace atom;

extern "C" {

int atom_kv_append_i4(void *kv_data, void *kv_param, const int32_t *kv_indptr, const int32_t *kv_indices,
                      const int32_t *last_page_offset, const void *k, const void *v, const void *k_param,
                      const void *v_param, const int32_t *append_indptr, int64_t total_tokens, int batch,
                      int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, void *stream) {
  const int st = check_kv(kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, batch, num_layers, layer_idx,
                          num_heads, page_size, head_dim);
  if (st != ATOM_OK) return st;
  if (!k || !v || !k_param || !v_param) return ATOM_ERR_INVALID_ARG;
  if (total_tokens < 1 || total_tokens > 0x7fffffff || (!append_indptr && total_tokens != batch)) return ATOM_ERR_SHAPE;
  if (!aligned16(k) || !aligned16(v) || (reinterpret_cast<uintptr_t>(k_param) & 3u) ||
      (reinterpret_cast<uintptr_t>(v_param) & 3u))
    return ATOM_ERR_ALIGN;
  AppendParams p{{(uint8_t *)kv_data, (half_t *)kv_param, kv_indptr, kv_indices, last_page_offset, batch, num_layers,
                  layer_idx, num_heads, page_size},
                 (const uint8_t *)k, (const uint8_t *)v, (const half_t *)k_param, (const half_t *)v_param,
                 append_indptr, total_tokens};
  hipLaunchKernelGGL(kv_append_kernel, dim3((unsigned)total_tokens), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  return check_launch();
}

int atom_kv_quant_append_f32(void *kv_data, void *kv_param, const int32_t *kv_indptr, const int32_t *kv_indices,
                             const int32_t *last_page_offset, const void *k_f32, const void *v_f32, int batch,
                             int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, void *stream) {
  const int st = check_kv(kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, batch, num_layers, layer_idx,
                          num_heads, page_size, head_dim);
  if (st != ATOM_OK) return st;
  if (!k_f32 || !v_f32) return ATOM_ERR_INVALID_ARG;
  if (!aligned16(k_f32) || !aligned16(v_f32)) return ATOM_ERR_ALIGN;
  QuantAppendParams p{{(uint8_t *)kv_data, (half_t *)kv_param, kv_indptr, kv_indices, last_page_offset, batch, num_layers,
                       layer_idx, num_heads, page_size},
                      (const float *)k_f32, (const float *)v_f32};
  hipLaunchKernelGGL(kv_quant_append_kernel, dim3((unsigned)(((int64_t)batch * 2 * num_heads + 7) / 8)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), p);
  return check_launch();
}

size_t atom_batch_decode_i4_workspace_bytes(int batch, int num_heads, int page_size, int max_pages_per_seq) {
  if (batch < 1 || num_heads < 1 || page_size < 16) return 0;
  const int s = decode_splits(batch, num_heads, max_pages_per_seq, page_size);
  return s > 1 ? (size_t)batch * num_heads * s * (kHeadDim + 2) * sizeof(float) : 0;
}

static int batch_decode_impl(void *o, const void *q, const float *k32, const float *v32, void *kv_data, void *kv_param,
                             const int32_t *kv_indptr, const int32_t *kv_indices, const int32_t *last_page_offset, int batch,
                             int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, float rope_theta,
                             float rope_scale, int max_pages_per_seq, void *workspace, size_t workspace_bytes, void *stream) {
  const int st = check_kv(kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, batch, num_layers, layer_idx,
                          num_heads, page_size, head_dim);
  if (st != ATOM_OK) return st;
  if (!q || !(rope_theta > 0.f) || !(rope_scale > 0.f)) return ATOM_ERR_INVALID_ARG;
  if ((o && !aligned16(o)) || !aligned16(q)) return ATOM_ERR_ALIGN;
  int total = decode_splits_total(batch, num_heads, max_pages_per_seq, page_size);     // waves per (sequence, head) pair
  // sequences x heads fill the chip on their own: the splits become the waves of one workgroup and merge in LDS -- one launch, no
  // workspace (same split count, same merge arithmetic: the same bits as kernel + decode_merge_kernel)
  const bool wgm = o && total > 1 && kWgmWaves % total == 0 && (int64_t)batch * num_heads >= ATOM_TUNE("ATOM_DECODE_WGM_PAIRS", 256);
  int inner = wgm ? 1 : decode_inner(batch, num_heads, total);
  int splits = total / inner;                           // partial states per pair (what the merge and the workspace see)
  const size_t need = (size_t)batch * num_heads * splits * (kHeadDim + 2) * sizeof(float);
  if (!wgm && splits > 1 && (!workspace || workspace_bytes < need || !aligned16(workspace))) { total = 1; inner = 1; splits = 1; }
  // o == NULL: the split partial states stay in the workspace un-merged (atom_batch_decode_i4_splits() of them; the consumer merges:
  // atom_gemm_w4a4_multi_merge_q) -- only meaningful when the KV range IS split
  if (!o && splits < 2) return ATOM_ERR_INVALID_ARG;
  DecodeParams p{{(uint8_t *)kv_data, (half_t *)kv_param, kv_indptr, kv_indices, last_page_offset, batch, num_layers,
                  layer_idx, num_heads, page_size},
                 make_fastdiv((unsigned)num_heads), make_fastdiv((unsigned)total), make_fastdiv((unsigned)(page_size >> 4)),
                 make_fastdiv((unsigned)page_size), wgm ? kWgmWaves / total : 1,
                 (const half_t *)q, (half_t *)o, (float *)workspace, total,
                 1.0f / sqrtf((float)kHeadDim), log2f(rope_theta), 1.0f / rope_scale, k32, v32};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (wgm) {
    const int ppw = kWgmWaves / total;                  // (sequence, head) pairs per workgroup
    hipLaunchKernelGGL(batch_decode_kernel<true>, dim3((unsigned)((batch * num_heads + ppw - 1) / ppw)), dim3(64 * kWgmWaves), 0, s, p);
    return check_launch();
  }
  if (inner == kInner)
    hipLaunchKernelGGL((batch_decode_kernel<false, kInner>), dim3((unsigned)(batch * num_heads), (unsigned)splits), dim3(64 * kInner), 0, s, p);
  else
    hipLaunchKernelGGL(batch_decode_kernel<false>, dim3((unsigned)(batch * num_heads), (unsigned)splits), dim3(64), 0, s, p);
  if (splits > 1 && o) {
    if (splits <= 8)
      hipLaunchKernelGGL(decode_merge_kernel<8>, dim3((unsigned)(batch * num_heads)), dim3(128), 0, s, (const float *)workspace, (half_t *)o, splits);
    else if (splits <= 16)
      hipLaunchKernelGGL(decode_merge_kernel<16>, dim3((unsigned)(batch * num_heads)), dim3(128), 0, s, (const float *)workspace, (half_t *)o, splits);
    else
      hipLaunchKernelGGL(decode_merge_kernel<32>, dim3((unsigned)(batch * num_heads)), dim3(128), 0, s, (const float *)workspace, (half_t *)o, splits);
  }
  return check_launch();
}

int atom_batch_decode_i4_splits(int batch, int num_heads, int page_size, int max_pages_per_seq) {
  if (batch < 1 || num_heads < 1 || page_size < 16) return 0;
  return decode_splits(batch, num_heads, max_pages_per_seq, page_size);
}

int atom_batch_decode_i4(void *o, const void *q, const void *kv_data, const void *kv_param, const int32_t *kv_indptr,
                         const int32_t *kv_indices, const int32_t *last_page_offset, int batch, int num_layers,
                         int layer_idx, int num_heads, int page_size, int head_dim, float rope_theta, float rope_scale,
                         int max_pages_per_seq, void *workspace, size_t workspace_bytes, void *stream) {
  return batch_decode_impl(o, q, nullptr, nullptr, const_cast<void *>(kv_data), const_cast<void *>(kv_param), kv_indptr, kv_indices, last_page_offset, batch, num_layers, layer_idx,
                           num_heads, page_size, head_dim, rope_theta, rope_scale, max_pages_per_seq, workspace, workspace_bytes, stream);
}

int atom_batch_decode_append_i4(void *o, const void *q, const void *k_f32, const void *v_f32, void *kv_data, void *kv_param,
                                const int32_t *kv_indptr, const int32_t *kv_indices, const int32_t *last_page_offset, int batch,
                                int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, float rope_theta,
                                float rope_scale, int max_pages_per_seq, void *workspace, size_t workspace_bytes, void *stream) {
  if (!k_f32 || !v_f32) return ATOM_ERR_INVALID_ARG;
  if (!aligned16(k_f32) || !aligned16(v_f32)) return ATOM_ERR_ALIGN;
  return batch_decode_impl(o, q, (const float *)k_f32, (const float *)v_f32, kv_data, kv_param, kv_indptr, kv_indices, last_page_offset,
                           batch, num_layers, layer_idx, num_heads, page_size, head_dim, rope_theta, rope_scale, max_pages_per_seq,
                           workspace, workspace_bytes, stream);
}

}  // extern "C"
