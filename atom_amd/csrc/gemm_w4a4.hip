// W4A4 group-128 mixed-precision GEMM for gfx950 (MI355X, CDNA4): C entry points and kernel dispatch.
//
//   D[M,N] = sum_g (A4_g . B4_g^T) sA[m,g] sB[g,n] + (A8 . B8^T) sA8[m] sB8[n]        (fp16 out)
//
// Replaces compute_gemm_imma / DenseLayerGEMM_i4_o16 (reference
// kernels/include/GEMM/Dense_layer_gemm_i4_o16.cuh:436-769).  NOT a translation: the reference is built on
// cp.async rings + ldmatrix + the native INT4 mma.m16n8k64; gfx950 has none of them.  The kernels:
//   gemm_w4a4_f6.hip   prefill, both operands BF6-coded on the block-scaled MFMA (ATOM_AB_F6; the fastest path)
//   gemm_w4a4_v3.hip   prefill, INT8 MFMA tiles (256x256 ... 64x64), packed or pre-widened (ATOM_A_WIDE) activations,
//                      split-K; the reference operand format
//   gemv_w4a4.hip      decode (M <= 7): weight streaming, v_dot8_i32_i4 on the packed dwords
//   gemm_w4a4_v2.hip   the 256x256 kernel with the asymmetric-u4 output epilogue (atom_gemm_w4a4_o4) and the
//                      ablation switches behind profiles/r01_ablation_v1_v2.txt
// Common design points: weights are the MFMA "A" operand and activations "B" (the accumulator tile is transposed, so a
// lane owns ONE token m and 16 features n: the token scale is a lane scalar); per-group de-quantisation is the reference's shape
// (Dense_layer_gemm_i4_o16.cuh:413-431) -- s = sA * sB, exact in FP32; c = fma(idot, s, c), ONE rounding per group -- which is 1.5 VALU
// per accumulator element in the BF6 headline kernel when the two channels of a weight_channel_group = 2 pair share the product
// (ATOM_B_SCALE_PAIRS), 2 otherwise; the INT8 kernels start each group's integer accumulator at the bit pattern of 1.5 * 2^23 so that
// the register READ AS A FLOAT is 12582912 + idot (one exact v_sub instead of an int->float conversion); the 128 INT8 keeper columns
// run as two extra 64-wide steps through the same pipeline.
#include "common.h"
#include <cstdlib>

using namespace atom;

static int fill_params(GemmParams &p, const void *A4, const void *B4, const void *sA, const void *sB, const void *A8,
                       const void *B8, const void *sA8, const void *sB8, int64_t M, int64_t N, int64_t K_total, int group,
                       int keeper, int scale_layout);

// largest M served by the weight-streaming decode kernel (ATOM_GEMV_MAXM overrides it for tuning)
static int gemv_max_m() { return ATOM_TUNE("ATOM_GEMV_MAXM", 7); }
// Up to this many tokens EVERY GEMM entry point (fp16, FP32 sums, segmented) runs the few-token dot-product kernel (gemv_w4a4.hip
// gemv1_w4a4_kernel: the weights streamed once at full occupancy, each token's sum in the one-token kernel's order); above it the
// MFMA decode-batch kernel.  One token always; two tokens where K is long (measured cold, us, dot-product | decode-batch kernel:
// 2 x 5120 x 13824 12.5 | 17.5, 2 x 4096 x 11008 8.0 | 8.6, 2 x 5120 x 5120 6.6 | 7.0, 2 x 4096 x 4096 4.6 | 4.4; from three tokens
// the per-token VALU work loses everywhere: 3 x 13824 x 5120 13.9 | 11.7; profiles/r04/decode_small_m.txt).  The rule depends on
// (M, K) only, so the projections that share an activation take the same kernel through every entry point.  (Round 6 tried two tokens
// on the dot-product kernel at every K, for the sake of the quantiser-in-front launch of gemvq_w4a4.hip: a Llama-7B layer at batch 2
// then takes 69-71 us cold against 66.6 with this rule and separate quantiser launches -- two tokens double the kernel's VALU work per
// weight chunk -- so the rule stayed and the decode layer fuses its quantisers at ONE token only.)
static int gemv_tokens(int64_t K_total) {
  const int forced = ATOM_TUNE("ATOM_GEMV_TOKENS", 0);       // (tuning builds)
  const int t = forced > 0 ? forced : (K_total > 4096 ? 2 : 1);
  return t > kGemvMaxTokens ? kGemvMaxTokens : t;
}

// Decode batches go to the register-resident weight-streaming MFMA kernel (gemm_w4a4_skinny.hip) where it measures
// faster than the tile kernels + split-K (profiles/r01_skinny.txt): always up to 16 tokens, up to 32 unless K is very
// long, up to 128 (256 for K <= 4096) while the shape is small enough that its one-workgroup-per-16-features grid and
// the per-workgroup re-widening of the activations are not the bottleneck.
static int skinny_max_m() { return ATOM_TUNE("ATOM_SKINNY_MAXM", 256); }
static bool skinny_fits(int64_t M, int64_t N, int64_t K_total) {
  const int64_t items = (K_total - kKeeper) / kGroup + 1;
  if (M > skinny_max_m() || items > 8 * 14) return false;
  if (M <= 16) return true;
  if (M <= 32) return items <= 96;
  if (M > 64 && items > 64) return false;                  // 8 / 16 token blocks: the 4- and 8-slot instances only
  if (M > 128) return items <= 32 && N * items <= 200000;   // 16 blocks: K <= 4096 (256x4096x4096: 16.1 vs 25.2 us)
  return N * items <= 420000;
}

// Mid-size batches in the packed (reference) format: 64x64 tiles over the whole K range on a deep LDS ring (gemm_w4a4_mid.hip, the INT8
// form).  Its K step costs ~120 instructions per wave -- widening nibbles, converting integers -- against ~60 of the BF6 form, so it
// only takes what neither the decode-batch kernel (skinny_fits: it wins wherever it applies, profiles/r05/mid_ab.txt) nor a cheap
// re-coding reaches: up to 256 rows, at most one tile per CU, K up to 11,264 (64 x 13824 x 5120: 27.3 -> 20.0 us, 256 x 4096 x 11008:
// 48.0 -> 38.9 us without a workspace; at 64 x 5120 x 13824 the split-K tiles stay ahead, 32.5 vs 47.6).
static bool skinny_fits(int64_t M, int64_t N, int64_t K_total);
static bool mid_fits(int64_t M, int64_t N, int64_t K_total) {
  if (!ATOM_TUNE("ATOM_MID", 1) || (N % 64) != 0) return false;
  const int64_t tiles = ((M + 63) / 64) * (N / 64), items = (K_total - kKeeper) / kGroup + 1;
  if (ATOM_TUNE("ATOM_MID_MIN_TILES", 0)) return tiles >= ATOM_TUNE("ATOM_MID_MIN_TILES", 0) && M <= ATOM_TUNE("ATOM_MID_MAX_M", 1024);   // (tuning builds)
  return M > 16 && M <= 256 && tiles >= 96 && tiles <= 256 && items <= 88 && !skinny_fits(M, N, K_total);
}

// Tile geometry of the F6 kernels by shape (measured: profiles/r02_f6_dispatch.txt, profiles/r03_f6_dispatch.txt -- every geometry forced
// on a 48-shape grid: this pick is within 1-3 % of the best one).  256x256 (one workgroup per CU, the q kernel) from 129 tiles in one
// round, or when its tiles keep >= 60 % of the CU slots of the rounds they need busy -- a full 256x256 tile does four 128x128 tiles'
// work in ~2.4x their time; K-group kernels (two / four groups of 4 waves per tile) while a shape yields at most 256 tiles of
// 128x128 / 64x128; the 256x128 kernel for 129..256 of its tiles; else 128x128 with 4 waves (several workgroups per CU).
static int f6_pick_cfg(int64_t M, int64_t N, int64_t K_total) {
  const int force = ATOM_TUNE("ATOM_F6_CFG", -1);
  if (force >= 0) return force;
  const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256), t128 = ((M + 127) / 128) * ((N + 127) / 128);
  const int64_t t64 = ((M + 63) / 64) * ((N + 127) / 128);
  const int64_t rounds = (t256 + 255) / 256, steps = (K_total - kKeeper) / kGroup + 2;
  // up to two 64x64 tiles per CU: the mid-size-batch kernel (gemm_w4a4_mid.hip; needs the float32 weight scales, else launch_gemm_f6
  // falls back to the 128x128 geometry -- the same K order).  Same box, us, K-group / 128x128 kernels -> this one
  // (profiles/r05/mid_f6c.txt): 64 x 4096 x 4096 13.6 -> 9.9, 256 x 4096 x 4096 14.3 -> 10.6, 512 x .. 16.9 -> 16.4, 64 x 13824 x 5120
  // 19.1 -> 12.2, 64 x 5120 x 13824 40.5 -> 24.6, 256 x 4096 x 11008 33.7 -> 22.9, 256 x 5120 x 5120 18.2 -> 17.3; beyond 512 tiles the
  // larger tiles win (1024 x 4096 x 4096 22.5 vs 29.2, 256 x 13824 x 5120 25.7 vs 33.6, 256 x 11008 x 4096 21.0 vs 23.2)
  if ((N % 64) == 0 && ((M + 63) / 64) * (N / 64) <= 512 && ATOM_TUNE("ATOM_F6_MID", 1)) return 20;
  // (one round: from 129 tiles -- 768x11008x4096, 129 tiles: 47.7 us against 51.6 on 128x128 tiles; with 128 or fewer the 256x128
  // kernel below has a tile for every CU.  Several rounds: only while >= 60 % of the slots are busy.  profiles/r03_f6_dispatch.txt)
  if (t256 >= 129 && (rounds == 1 || 5 * t256 >= 3 * rounds * 256)) return 0;
  // at most one tile per CU: a lone 4-wave workgroup is latency-bound (barrier, fragment loads: ~1 us per K step), so two
  // groups of 4 waves share the tile and its K steps (profiles/r02_mid_m.txt: 1024x4096x4096 33.1 -> 23.4 us, 512x..: 26.3 ->
  // 18.4).  The result is the sum of two (four) ordered ranges of the K steps (atom_gemm_w4a4_f6_order).
#ifdef ATOM_TOOLS   // (rounds 2-4: up to 256 tiles of 64x128 four / two K groups shared the tile; the mid-size-batch kernel above takes those shapes now)
  if (steps >= 16 && t64 <= 256) return 12;               // ... four groups on a 64x128 tile (K = 4096: 17.9 -> 15.0 us at 256 rows)
  if (steps >= 8 && t64 <= 256) return 9;
#else
  (void)t64;
#endif
  if (steps >= 8 && t128 <= 256) return 6;
  // more 128x128 tiles than CUs, but at most one 256x128 tile per CU and more than half of them busy: the 256x128 q-step kernel
  // (ATOM_B_F6S weights; launch_gemm_f6 runs the 128x128 geometry otherwise -- same K order)
  const int64_t t2 = ((M + 255) / 256) * ((N + 127) / 128);
  if (steps >= 6 && t2 > 128 && t2 <= 256 && ATOM_TUNE("ATOM_Q2", 1)) return 8;
  return 3;
}

// ATOM_WS_VERIFY (debug calls of atom_gemm_w4a4_f16_ws): CHECK the caller's assertions before using them -- a wrong ATOM_B_SCALE_PAIRS or
// ATOM_WS_WEIGHT_CACHED otherwise gives wrong numbers without any error.  Violations are counted on the device into the last 16 bytes of
// the workspace (free until the GEMM's own use of it starts), copied back, and the stream is SYNCHRONISED: not for production calls,
// not during graph capture.
static bool f6_route(int64_t M, int64_t N, int64_t K_total);
static bool f6_route_cached(int64_t M, int64_t N, int64_t K_total);
static int verify_assertions(const GemmParams &p, int64_t M, int64_t N, int64_t K_total, int flags, void *workspace, size_t workspace_bytes,
                             hipStream_t hs) {
  if (!workspace || workspace_bytes < 16 || !aligned16(workspace)) return ATOM_ERR_INVALID_ARG;
  int32_t *cnt = reinterpret_cast<int32_t *>((uint8_t *)workspace + ((workspace_bytes - 16) & ~(size_t)15));
  if (hipMemsetAsync(cnt, 0, 16, hs) != hipSuccess) return ATOM_ERR_LAUNCH;
  if (p.b_pairs && !p.f6_rows_a) {
    const int r = launch_check_scale_pairs(p.sB, p.G, N, cnt, hs);
    if (r != ATOM_OK) return r;
  }
  if ((flags & ATOM_WS_WEIGHT_CACHED) && !p.a_wide && !p.f6_rows_a && (f6_route(M, N, K_total) || f6_route_cached(M, N, K_total))) {
    if (workspace_bytes < atom_gemm_w4a4_workspace_bytes(M, N, K_total)) return ATOM_ERR_INVALID_ARG;   // no weight region to speak of
    const int r = launch_verify_weight_f6s(p.B4, p.sB, N, p.K4h, p.G, (const uint8_t *)workspace, cnt + 1, hs);
    if (r != ATOM_OK) return r;
  }
  int32_t host[2] = {0, 0};
  if (hipMemcpyAsync(host, cnt, sizeof(host), hipMemcpyDeviceToHost, hs) != hipSuccess || hipStreamSynchronize(hs) != hipSuccess)
    return ATOM_ERR_LAUNCH;
  return host[0] != 0 || host[1] != 0 ? ATOM_ERR_INVALID_ARG : ATOM_OK;
}

extern "C" {

int atom_gemm_w4a4_f6_order(int64_t M, int64_t N, int64_t K_total) {
  if (M < 1 || N < 64 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0) return 0;
  const int cfg = f6_pick_cfg(M, N, K_total);
  return cfg == 12 ? 4 : ((cfg == 5 || cfg == 6 || cfg == 9) ? 2 : 1);   // (12 / 9 / 5: tuning builds only)
}

const char *atom_version(void) { return "atom_hip 0.1 (gfx950)"; }

const char *atom_strerror(int code) {
  switch (code) {
    case ATOM_OK: return "ok";
    case ATOM_ERR_INVALID_ARG: return "invalid argument (null pointer or bad enum)";
    case ATOM_ERR_SHAPE: return "unsupported shape / group / keeper";
    case ATOM_ERR_ALIGN: return "pointer not 16-byte aligned";
    case ATOM_ERR_LAUNCH: return "HIP launch failed";
    default: return "unknown error";
  }
}

int atom_gemm_w4a4_f16(const void *A4, const void *B4, const void *sA, const void *sB, const void *A8,
                       const void *B8, const void *sA8, const void *sB8, void *D, int64_t M, int64_t N,
                       int64_t K_total, int group, int keeper, int scale_layout, void *stream) {
  if (!D) return ATOM_ERR_INVALID_ARG;
  GemmParams p;
  const int fst = fill_params(p, A4, B4, sA, sB, A8, B8, sA8, sB8, M, N, K_total, group, keeper, scale_layout);
  if (fst != ATOM_OK) return fst;
  if (!aligned16(D)) return ATOM_ERR_ALIGN;
  p.D = (half_t *)D;
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
#ifdef ATOM_TOOLS
  const int variant = ATOM_TUNE("ATOM_GEMM_VARIANT", 0);   // tuning / ablation variants (tools build only); 0 is the product path
  if (p.a_wide && variant != 0 && !(variant >= 320 && variant <= 330)) return ATOM_ERR_INVALID_ARG;
  if (p.f6_rows_a && variant != 0) return ATOM_ERR_INVALID_ARG;
  switch (variant) {
    case 320: case 324: case 325: return launch_gemm_v3(p, variant - 300, hs);
    case 203: return launch_gemm_v2(p, 3, hs);
    case 204: return launch_gemm_v2(p, 4, hs);
    case 1001: case 1002: case 1003: case 1004: case 1008: case 1016: case 1019: case 1023: case 1031: case 1032: case 1033: case 1035: case 1064: case 1128: case 1256:
      return launch_gemm_v2(p, variant, hs);
    case 2: return launch_gemm_v2(p, 4, hs);
    case 300: case 301: case 302: case 303: case 304: case 305: case 306: return launch_gemm_v3(p, variant - 300, hs);
    case 310: case 311: case 330: {   // traced run: the trace buffer pointer arrives in ATOM_TRACE_PTR (tools/trace_gemm.cpp)
      const char *e = getenv("ATOM_TRACE_PTR");
      if (!e) return ATOM_ERR_INVALID_ARG;
      p.Dsz = reinterpret_cast<half_t *>(strtoull(e, nullptr, 16));
      return launch_gemm_v3(p, variant - 300, hs);
    }
    default: break;
  }
#endif
  {
      if (p.f6_rows_a) return launch_gemm_f6(p, f6_pick_cfg(M, N, K_total), hs);   // BF6 operands: block-scaled MFMA kernels
      if (p.a_wide) {   // activations pre-widened by the quant kernels: 256x256 tiles once they fill half the chip
        const int64_t cm256 = (M + 255) / 256, cn256 = (N + 255) / 256, t5 = ((M + 63) / 64) * ((N + 127) / 128);
        const int cfg = cm256 * cn256 >= 128 ? 20 : ((t5 > 256 && t5 < 1024) ? 24 : 25);
        return launch_gemm_v3(p, cfg, hs);
      }
      if (M <= gemv_tokens(K_total)) {                                     // a few tokens: the dot-product weight stream
        const int st = launch_gemv1(p, hs);
        if (st != ATOM_ERR_SHAPE) return st;
      }
      if (mid_fits(M, N, K_total)) {                                // mid-size batches: 64x64 tiles on a deep LDS ring
        const int st = launch_gemm_mid(p, hs);
        if (st != ATOM_ERR_SHAPE) return st;
      }
      if (M > 1 && skinny_fits(M, N, K_total)) {                    // decode batches: weight streaming on the MFMA
        const int st = launch_gemm_skinny(p, hs);
        if (st != ATOM_ERR_SHAPE) return st;
      }
      if (M <= gemv_max_m()) {                                      // M = 1 (and 2..7 with K too long for the above):
        const int st = launch_gemv(p, hs);                          // weight-streaming dot-product kernel
        if (st != ATOM_ERR_SHAPE) return st;
      }
      {   // prefill: LDS-DMA MFMA tile kernel (gemm_w4a4_v3.hip); tile geometry by how many workgroups the shape yields
          // (measured on MI355X, profiles/r01_tile_selection.txt): big tiles only once they fill the chip.
        const int64_t cm256 = (M + 255) / 256, cm64 = (M + 63) / 64, cn256 = (N + 255) / 256, cn128 = (N + 127) / 128;
        int cfg;
        if (cm256 * cn256 >= 1024) cfg = 0;               // 256x256, 8 waves, 1 WG/CU
        else if (cm256 * cn128 >= 512) cfg = 1;           // 256x128, 4 waves, 2 WGs/CU
        else {
          const int64_t t5 = cm64 * cn128;                // 64x128, 2 waves
          cfg = (t5 > 256 && t5 < 1024) ? 4 : 5;          // in between: 64x64, 1 wave, twice the workgroups
        }
        return launch_gemm_v3(p, cfg, hs);
      }
  }
}

static int fill_params(GemmParams &p, const void *A4, const void *B4, const void *sA, const void *sB, const void *A8,
                       const void *B8, const void *sA8, const void *sB8, int64_t M, int64_t N, int64_t K_total, int group,
                       int keeper, int scale_layout) {
  if (!A4 || !B4 || !sA || !sB || !A8 || !B8 || !sA8 || !sB8) return ATOM_ERR_INVALID_ARG;
  const int a_wide = (scale_layout & ATOM_A_WIDE) != 0;
  const int f6 = (scale_layout & ATOM_AB_F6) != 0, f6s = (scale_layout & ATOM_B_F6S) != 0;
  p.o4_ref = (scale_layout & ATOM_O4_REF_EXTREMA) != 0;    // (only the _o4 entry points look at it)
  p.b_pairs = (scale_layout & ATOM_B_SCALE_PAIRS) != 0;
  scale_layout &= ~(ATOM_A_WIDE | ATOM_AB_F6 | ATOM_B_F6S | ATOM_O4_REF_EXTREMA | ATOM_WS_WEIGHT_CACHED | ATOM_B_SCALE_PAIRS | ATOM_WS_VERIFY);
  if ((a_wide && f6) || (f6s && !f6)) return ATOM_ERR_INVALID_ARG;
  if (scale_layout != ATOM_SCALE_LAYOUT_REF && scale_layout != ATOM_SCALE_LAYOUT_PLAIN) return ATOM_ERR_INVALID_ARG;
  if (group != kGroup || keeper != kKeeper) return ATOM_ERR_SHAPE;
  if (M < 1 || N < 64 || (N % 64) != 0 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0) return ATOM_ERR_SHAPE;
  if (M > (1 << 24) || N > (1 << 24) || K_total > (1 << 20)) return ATOM_ERR_SHAPE;
  if ((M > N ? M : N) * ((K_total - kKeeper) / 2) >= (int64_t(1) << 32)) return ATOM_ERR_SHAPE;   // 32-bit DMA offsets
  if (a_wide && M * (K_total - kKeeper) >= (int64_t(1) << 32)) return ATOM_ERR_SHAPE;
  if (!aligned16(A4) || !aligned16(B4) || !aligned16(A8) || !aligned16(B8)) return ATOM_ERR_ALIGN;
  if ((reinterpret_cast<uintptr_t>(sB) & 3u) || (reinterpret_cast<uintptr_t>(sB8) & 3u)) return ATOM_ERR_ALIGN;
  p.A4 = (const uint8_t *)A4; p.B4 = (const uint8_t *)B4;
  p.sA = (const half_t *)sA;  p.sB = (const half_t *)sB;
  p.A8 = (const uint8_t *)A8; p.B8 = (const uint8_t *)B8;
  p.sA8 = (const half_t *)sA8; p.sB8 = (const half_t *)sB8;
  p.D = nullptr; p.D4 = nullptr; p.Dsz = nullptr; p.ws = nullptr; p.splits = 1; p.q_op = 0; p.q_part = nullptr; p.q_splits = 0; p.q_roles = 0;
  p.M = (int)M; p.N = (int)N;
  p.K4h = (int)((K_total - kKeeper) / 2);
  p.G = (int)((K_total - kKeeper) / kGroup);
  p.ref_layout = scale_layout == ATOM_SCALE_LAYOUT_REF;
  p.a_wide = a_wide;
  p.f6_rows_a = f6 ? (M + 255) / 256 * 256 : 0;              // == atom_f6_rows(): rows per group, padded to the tile
  p.f6_rows_b = f6 ? (N + 255) / 256 * 256 : 0;
  p.sB32 = f6s ? reinterpret_cast<const float *>((const uint8_t *)B4 + (size_t)p.G * (size_t)p.f6_rows_b * 104) : nullptr;
  p.ldA = (int64_t)atom_scale_size(M, scale_layout);
  return ATOM_OK;
}

// Split-K policy: shapes that yield fewer than 512 workgroups of the smallest tile are latency-bound (one pass over K per
// workgroup at ~1 us per K-group); split the K loop over up to 8 workgroups and reduce FP32 partials in a second launch.
// `packed`: the operands are packed nibbles (pre-widened ATOM_A_WIDE activations do not reach the mid-size-batch kernel, so its shapes
// keep their split-K route for those)
static int choose_splits(int64_t M, int64_t N, int64_t K_total, bool packed = true) {
  if (M <= gemv_max_m() || skinny_fits(M, N, K_total) || (packed && mid_fits(M, N, K_total))) return 1;   // decode / mid-size kernels
  const int64_t tiles = ((M + 63) / 64) * ((N + 127) / 128);
  const int64_t nsteps = (K_total - kKeeper) / kGroup + 2;
  const int force = ATOM_TUNE("ATOM_SPLITS", 0);
  if (force > 0) return force > nsteps / 2 ? (int)(nsteps / 2) : force;
  if (tiles >= 512 || nsteps < 8) return 1;
  int64_t s = 1024 / tiles;                                // measured (profiles/r01_gemm_sweeps.txt): 512x4096x4096 1 -> 4
  if (s > 8) s = 8;                                        // splits: 38.4 -> 31.3 us; 256x13824x5120 1 -> 2: 51.3 -> 47.2 us
  if (tiles >= 96 && s > 4) s = 4;
  if (s > nsteps / 4) s = nsteps / 4;
  return s < 2 ? 1 : (int)s;
}

// Packed operands of prefill size: re-code both into the F6 format (one bandwidth-bound launch, ~12 us with its launch gap at
// N = K = 4096; the activation alone, 3-6 us, once the weight's form is cached in the caller's workspace: ATOM_WS_WEIGHT_CACHED) and
// run the block-scaled-MFMA kernels: 59-68 instead of 106 us at 4096^3.  From 257 rows, and from 129 where the decode-batch kernel
// does not take the shape (profiles/r03_packed_route.txt, INT8 kernels | route, weight cached | not cached, us: 384x4096x4096 28.5 |
// 20.6 | 27.6, 512x.. 29.6 | 21.6 | 29.1, 768x.. 37.4 | 26.3 | 32.3, 512x11008x4096 49.9 | 35.5 | 44.3, 256x11008x4096 36.2 | 25.3 |
// 34.8; at 256x4096x4096 the decode-batch kernel's 15.8 stands against 19.6 | 25.3).  Round 2 drew the line at 768 rows, with the
// weight re-coded by every call.
static bool f6_route(int64_t M, int64_t N, int64_t K_total) {
  const int off = ATOM_TUNE("ATOM_NO_F6_ROUTE", 0);
  if (off || N < 2048 || K_total < 1024 || mid_fits(M, N, K_total)) return false;
  if (M >= ATOM_TUNE("ATOM_F6_ROUTE_MIN_M", 257)) return true;
  return M > 128 && !skinny_fits(M, N, K_total);
}
// ... and with the weight's BF6 form already in the workspace (ATOM_WS_WEIGHT_CACHED) from 129 rows whatever the decode kernels take: only
// the activation is re-coded, and the mid-size-batch kernel runs 129 .. 256 rows in ~10 us at 4096 x 4096 (decode-batch kernel: 9.4 .. 15.5)
// -- and from 17 rows where the decode-batch kernel does not take the shape (large N x K): re-coding 64 activation rows costs ~2.5 us and
// the BF6 mid-size-batch kernel then runs 64 x 13824 x 5120 in 12.2 us where the INT8 form takes 22.4, 64 x 5120 x 13824 in 24.6 against
// 32.5 on split-K tiles (profiles/r05/mid_f6c.txt, bench.py configs rows)
static bool f6_route_cached(int64_t M, int64_t N, int64_t K_total) {
  if (ATOM_TUNE("ATOM_NO_F6_ROUTE", 0) || N < 2048 || K_total < 1024) return false;
  return M > 128 || (M > 16 && !skinny_fits(M, N, K_total));
}
static size_t f6_bytes(int64_t rows, int64_t K_total) {
  return (size_t)((K_total - kKeeper) / kGroup) * (size_t)((rows + 255) / 256 * 256) * 104;
}

int atom_gemm_w4a4_packed_order(int64_t M, int64_t N, int64_t K_total, int with_workspace) {
  if (M < 1 || N < 64 || (N % 64) != 0 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0) return 0;
  if (with_workspace && atom_gemm_w4a4_workspace_bytes(M, N, K_total) != 0) {
    if (f6_route(M, N, K_total) || (with_workspace == 2 && f6_route_cached(M, N, K_total)))
      return atom_gemm_w4a4_f6_order(M, N, K_total);                                 // re-coded to BF6: 1 / 2
    // split-K through the workspace -- never with ATOM_WS_WEIGHT_CACHED: the workspace's head holds the weight's BF6 form then, and
    // FP32 partial sums written there would destroy it (atom_gemm_w4a4_f16_ws runs such a call as the plain entry point does)
    if (with_workspace != 2 && choose_splits(M, N, K_total) > 1) return 100 + choose_splits(M, N, K_total);
  }
  if (M <= gemv_tokens(K_total)) return 64;                                          // the dot-product kernel
  if (mid_fits(M, N, K_total)) return 1;
  if (M > 1 && skinny_fits(M, N, K_total)) return 8;                                 // the decode-batch kernel
  if (M <= gemv_max_m()) return 63;                                                  // the staged dot-product kernel
  return 1;                                                                          // tile kernels
}

int atom_gemm_w4a4_ws_recodes(int64_t M, int64_t N, int64_t K_total) {
  if (M < 1 || N < 64 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0) return 0;
  return f6_route(M, N, K_total) ? 1 : 0;
}

int atom_gemm_w4a4_ws_recodes_cached(int64_t M, int64_t N, int64_t K_total) {
  if (M < 1 || N < 64 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0) return 0;
  return f6_route(M, N, K_total) || f6_route_cached(M, N, K_total) ? 1 : 0;
}

size_t atom_gemm_w4a4_workspace_bytes(int64_t M, int64_t N, int64_t K_total) {
  if (M < 1 || N < 64 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0) return 0;
  const int s = choose_splits(M, N, K_total, false);                           // (the larger of the two operand formats' needs)
  const size_t split = s > 1 ? (size_t)s * (size_t)M * (size_t)N * sizeof(float) : 0;
  const size_t f6 = f6_bytes(M, K_total) + f6_bytes(N, K_total) / 104 * 108;     // B: + float32 scales
  if (f6_route(M, N, K_total)) return f6;
  if (f6_route_cached(M, N, K_total)) return f6 > split ? f6 : split;             // (splits K without ATOM_WS_WEIGHT_CACHED, re-codes with it)
  return split;
}

int atom_gemm_w4a4_f16_ws(const void *A4, const void *B4, const void *sA, const void *sB, const void *A8, const void *B8,
                          const void *sA8, const void *sB8, void *D, int64_t M, int64_t N, int64_t K_total, int group,
                          int keeper, int scale_layout, void *workspace, size_t workspace_bytes, void *stream) {
  const size_t need = atom_gemm_w4a4_workspace_bytes(M, N, K_total);
  if (scale_layout & ATOM_WS_VERIFY) {                       // debug call: the caller's assertions are checked first (synchronises)
    GemmParams pv;
    const int fv = fill_params(pv, A4, B4, sA, sB, A8, B8, sA8, sB8, M, N, K_total, group, keeper, scale_layout);
    if (fv != ATOM_OK) return fv;
    const int rv = verify_assertions(pv, M, N, K_total, scale_layout, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
    if (rv != ATOM_OK) return rv;
  }
  if (need == 0 || !workspace || workspace_bytes < need)
    return atom_gemm_w4a4_f16(A4, B4, sA, sB, A8, B8, sA8, sB8, D, M, N, K_total, group, keeper, scale_layout, stream);
  if (!D) return ATOM_ERR_INVALID_ARG;
  GemmParams p;
  const int st = fill_params(p, A4, B4, sA, sB, A8, B8, sA8, sB8, M, N, K_total, group, keeper, scale_layout);
  if (st != ATOM_OK) return st;
  if (!aligned16(D) || !aligned16(workspace) || (N % 8) != 0) return ATOM_ERR_ALIGN;
  p.D = (half_t *)D;
  const bool wcached = (scale_layout & ATOM_WS_WEIGHT_CACHED) != 0;
  // ATOM_WS_WEIGHT_CACHED: the head of the workspace is the weight's BF6 form, valid across calls of ANY batch size -- a call that does
  // not take the re-coding route must not use the workspace for anything else (round 5 wrote split-K partial sums over it: 8 .. 16 rows
  // at K_total > 14464, e.g. Llama-70B down_proj; the next call from 17 rows then multiplied garbage).  Such a call runs as the plain
  // entry point does: atom_gemm_w4a4_packed_order(.., 2) says so.
  if (wcached && !p.a_wide && !p.f6_rows_a && !f6_route(M, N, K_total) && !f6_route_cached(M, N, K_total))
    return atom_gemm_w4a4_f16(A4, B4, sA, sB, A8, B8, sA8, sB8, D, M, N, K_total, group, keeper, scale_layout, stream);
  if (!p.a_wide && !p.f6_rows_a && !f6_route(M, N, K_total) && !(wcached && f6_route_cached(M, N, K_total)) && choose_splits(M, N, K_total) <= 1)
    return atom_gemm_w4a4_f16(A4, B4, sA, sB, A8, B8, sA8, sB8, D, M, N, K_total, group, keeper, scale_layout, stream);   // (129 .. 256 rows, nothing cached)
  if (!p.a_wide && !p.f6_rows_a && (f6_route(M, N, K_total) || (wcached && f6_route_cached(M, N, K_total)))) {   // packed operands -> F6 copies in the workspace
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    // layout: weight records, their float32 scales, then the activation records -- the weight region does not move with M, so a
    // weight re-coded once serves later calls of any batch size (ATOM_WS_WEIGHT_CACHED)
    uint8_t *b6 = (uint8_t *)workspace;
    float *sb32 = reinterpret_cast<float *>(b6 + f6_bytes(N, K_total));
    uint8_t *a6 = b6 + f6_bytes(N, K_total) / 104 * 108;
    // ATOM_WS_WEIGHT_CACHED: b6 / sb32 hold this weight's F6 form since an earlier call (the caller's assertion): activation only
    const int r = wcached
                      ? launch_repack_f6(p.A4, M, p.K4h, p.G, p.sA, p.ldA, p.ref_layout, a6, hs)
                      : launch_repack_f6_pair(p.A4, M, p.sA, p.ldA, p.ref_layout, a6, p.sB, sb32, p.B4, N, b6, p.K4h, p.G, hs);
    if (r != ATOM_OK) return r;
    p.A4 = a6; p.B4 = b6;
    p.f6_rows_a = (M + 255) / 256 * 256;
    p.f6_rows_b = (N + 255) / 256 * 256;
    p.sB32 = sb32;
    return launch_gemm_f6(p, f6_pick_cfg(M, N, K_total), hs);
  }
  if (p.a_wide || p.f6_rows_a) {                                      // native formats: no workspace route for these sizes
    if (choose_splits(M, N, K_total, !p.a_wide) <= 1)
      return atom_gemm_w4a4_f16(A4, B4, sA, sB, A8, B8, sA8, sB8, D, M, N, K_total, group, keeper, scale_layout, stream);
  }
  p.ws = (float *)workspace;
  p.splits = choose_splits(M, N, K_total, !p.a_wide);
  if (p.f6_rows_a) {                                                  // the F6 kernels need no workspace (tools: ATOM_F6_SPLITS3)
    const int cfg = f6_pick_cfg(M, N, K_total);
    const int sp = ATOM_TUNE("ATOM_F6_SPLITS3", 0);
    if ((cfg == 3 || cfg == 4) && sp > 1 && sp <= p.splits) p.splits = sp;
    else { p.ws = nullptr; p.splits = 1; }
    return launch_gemm_f6(p, cfg, reinterpret_cast<hipStream_t>(stream));
  }
  return launch_gemm_v3(p, p.a_wide ? 25 : 5, reinterpret_cast<hipStream_t>(stream));
}

int atom_gemm_w4a4_o4(const void *A4, const void *B4, const void *sA, const void *sB, const void *A8, const void *B8,
                      const void *sA8, const void *sB8, void *D_u4, void *D_scale_zero, int64_t M, int64_t N,
                      int64_t K_total, int group, int keeper, int scale_layout, void *stream) {
  if (!D_u4 || !D_scale_zero) return ATOM_ERR_INVALID_ARG;
  GemmParams p;
  const int st = fill_params(p, A4, B4, sA, sB, A8, B8, sA8, sB8, M, N, K_total, group, keeper, scale_layout);
  if (st != ATOM_OK) return st;
  if ((N % 128) != 0) return ATOM_ERR_SHAPE;
  if (p.a_wide || p.f6_rows_a) return ATOM_ERR_INVALID_ARG;  // the u4 epilogue kernel takes packed activations only
  if (!aligned16(D_u4)) return ATOM_ERR_ALIGN;
  p.D4 = (uint8_t *)D_u4;
  p.Dsz = (half_t *)D_scale_zero;
  return launch_gemm_v2_o4(p, reinterpret_cast<hipStream_t>(stream));
}

int atom_gemm_w4a4_f32(const void *A4, const void *B4, const void *sA, const void *sB, const void *A8, const void *B8,
                       const void *sA8, const void *sB8, void *D_f32, int64_t M, int64_t N, int64_t K_total, int group,
                       int keeper, int scale_layout, void *stream) {
  if (!D_f32) return ATOM_ERR_INVALID_ARG;
  if (scale_layout & (ATOM_A_WIDE | ATOM_AB_F6)) return ATOM_ERR_INVALID_ARG;
  GemmParams p;
  const int st = fill_params(p, A4, B4, sA, sB, A8, B8, sA8, sB8, M, N, K_total, group, keeper, scale_layout);
  if (st != ATOM_OK) return st;
  if (!aligned16(D_f32)) return ATOM_ERR_ALIGN;
  if (!skinny_fits(M, N, K_total)) return ATOM_ERR_SHAPE;
  p.ws = (float *)D_f32;
  if (M <= gemv_tokens(K_total)) return launch_gemv1_f32(p, reinterpret_cast<hipStream_t>(stream));   // a few tokens: the dot-product kernel (and ITS summation order) behind every entry point
  return launch_gemm_skinny_f32(p, reinterpret_cast<hipStream_t>(stream));
}

int atom_gemm_w4a4_multi_fits(int64_t M, int64_t N_seg, int nseg, int64_t K_total) {
  if (M < 1 || nseg < 1 || nseg > 3 || N_seg < 16 || (N_seg % 16) != 0 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0) return 0;
  const int64_t items = (K_total - kKeeper) / kGroup + 1;
  if (items > 64 && M > 16) return 0;                      // 14-group slices per wave: one token block (the larger instances spill)
  return skinny_fits(M, N_seg * nseg, K_total) && items <= 8 * 14 ? 1 : 0;
}

int atom_gemm_w4a4_multi(const void *A4, const void *B4, const void *sA, const void *sB, const void *A8, const void *B8,
                         const void *sA8, const void *sB8, void *out0, void *out1, void *out2, unsigned f32_mask,
                         const void *add0_f16, int64_t M, int64_t N_seg, int nseg, int64_t K_total, int group, int keeper,
                         int scale_layout, void *stream) {
  if (nseg < 1 || nseg > 3 || !out0 || (nseg > 1 && !out1) || (nseg > 2 && !out2)) return ATOM_ERR_INVALID_ARG;
  if (scale_layout & (ATOM_A_WIDE | ATOM_AB_F6)) return ATOM_ERR_INVALID_ARG;
  if (N_seg < 16 || (N_seg % 16) != 0) return ATOM_ERR_SHAPE;
  if ((f32_mask & 1u) && add0_f16) return ATOM_ERR_INVALID_ARG;      // the addend goes with an fp16 segment 0
  GemmParams p;
  const int st = fill_params(p, A4, B4, sA, sB, A8, B8, sA8, sB8, M, N_seg * nseg, K_total, group, keeper, scale_layout);
  if (st != ATOM_OK) return st;
  if (!atom_gemm_w4a4_multi_fits(M, N_seg, nseg, K_total)) return ATOM_ERR_SHAPE;
  if (!aligned16(out0) || (out1 && !aligned16(out1)) || (out2 && !aligned16(out2)) || (add0_f16 && !aligned16(add0_f16))) return ATOM_ERR_ALIGN;
  p.seg_out[0] = out0; p.seg_out[1] = out1; p.seg_out[2] = out2;
  p.seg_add = (const half_t *)add0_f16;
  p.seg_n = (int)N_seg;
  p.seg_f32 = f32_mask;
  if (M <= gemv_tokens(K_total)) return launch_gemv1_multi(p, reinterpret_cast<hipStream_t>(stream));  // (see atom_gemm_w4a4_f32)
  return launch_gemm_skinny_multi(p, reinterpret_cast<hipStream_t>(stream));
}

// one or two tokens (the token counts of gemv_tokens(): the projections take the dot-product kernel through every entry point): the
// quantiser in front of THAT kernel, once per CU (gemvq_w4a4.hip, round 6); otherwise in front of the decode-batch kernel
static bool multi_q_dot(int q_op, int64_t M, int64_t N, int64_t K_total) {
  return ATOM_TUNE("ATOM_GEMVQ", 1) && M <= gemv_tokens(K_total) && gemvq_fits(q_op, M, N, K_total);
}

int atom_gemm_w4a4_multi_q_fits(int q_op, int64_t M, int64_t N_seg, int nseg, int64_t K_total) {
  if (M < 1 || nseg < 1 || nseg > 3 || N_seg < 16 || (N_seg % 16) != 0 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0) return 0;
  if (multi_q_dot(q_op, M, N_seg * nseg, K_total)) return atom_gemm_w4a4_multi_fits(M, N_seg, nseg, K_total);
  if (!skinny_q_fits(q_op, M, K_total)) return 0;                    // the launcher's own predicate (gemm_w4a4_skinny.hip)
  return atom_gemm_w4a4_multi_fits(M, N_seg, nseg, K_total);
}

int atom_gemm_w4a4_multi_q(int q_op, const void *x, const void *x2, const void *residual, void *residual_out,
                           const int16_t *reorder_index, float eps, float clip, const void *B4, const void *sB, const void *B8,
                           const void *sB8, void *out0, void *out1, void *out2, unsigned f32_mask, const void *add0_f16, int64_t M,
                           int64_t N_seg, int nseg, int64_t K_total, int group, int keeper, void *stream) {
  if (q_op < ATOM_Q_REORDER || q_op > ATOM_Q_SILU_MUL || !x) return ATOM_ERR_INVALID_ARG;
  if ((q_op == ATOM_Q_RMSNORM || q_op == ATOM_Q_ADD_RMSNORM || q_op == ATOM_Q_SILU_MUL) && !x2) return ATOM_ERR_INVALID_ARG;
  if (q_op == ATOM_Q_ADD_RMSNORM && (!residual || !residual_out || residual_out == residual || residual_out == x)) return ATOM_ERR_INVALID_ARG;
  if (q_op == ATOM_Q_SILU_MUL && reorder_index) return ATOM_ERR_INVALID_ARG;
  if (nseg < 1 || nseg > 3 || !out0 || (nseg > 1 && !out1) || (nseg > 2 && !out2)) return ATOM_ERR_INVALID_ARG;
  if (N_seg < 16 || (N_seg % 16) != 0) return ATOM_ERR_SHAPE;
  if ((f32_mask & 1u) && add0_f16) return ATOM_ERR_INVALID_ARG;
  if (!(clip > 0.f) || !(eps >= 0.f)) return ATOM_ERR_INVALID_ARG;
  GemmParams p;
  // (the packed activation operand does not exist: the kernel builds it in LDS; x / sB stand in for the pointer checks)
  const int st = fill_params(p, x, B4, sB, sB, x, B8, sB, sB8, M, N_seg * nseg, K_total, group, keeper, ATOM_SCALE_LAYOUT_PLAIN);
  if (st != ATOM_OK) return st;
  if (!atom_gemm_w4a4_multi_q_fits(q_op, M, N_seg, nseg, K_total)) return ATOM_ERR_SHAPE;
  if (!aligned16(out0) || (out1 && !aligned16(out1)) || (out2 && !aligned16(out2)) || (add0_f16 && !aligned16(add0_f16))) return ATOM_ERR_ALIGN;
  if (!aligned16(x) || (q_op == ATOM_Q_SILU_MUL && !aligned16(x2)) || (residual && !aligned16(residual)) ||
      (residual_out && !aligned16(residual_out)) || (reorder_index && !aligned16(reorder_index)))
    return ATOM_ERR_ALIGN;
  p.A4 = nullptr; p.sA = nullptr; p.A8 = nullptr; p.sA8 = nullptr;
  p.seg_out[0] = out0; p.seg_out[1] = out1; p.seg_out[2] = out2;
  p.seg_add = (const half_t *)add0_f16;
  p.seg_n = (int)N_seg;
  p.seg_f32 = f32_mask;
  p.q_op = q_op;
  p.q_x = (const half_t *)x; p.q_x2 = (const half_t *)x2;
  p.q_res = (const half_t *)residual; p.q_res_out = (half_t *)residual_out;
  p.q_idx = reorder_index;
  p.q_eps = eps; p.q_clip = clip;
#ifdef ATOM_TOOLS   // traced run (tools/r06/gemvq_trace.py): the stamp buffer arrives in ATOM_TRACE_PTR
  if (const char *e = getenv("ATOM_TRACE_PTR")) p.Dsz = reinterpret_cast<half_t *>(strtoull(e, nullptr, 16));
#endif
  if (multi_q_dot(q_op, M, N_seg * nseg, K_total)) return launch_gemvq_multi_q(p, reinterpret_cast<hipStream_t>(stream));
  return launch_gemm_skinny_multi_q(p, reinterpret_cast<hipStream_t>(stream));
}

int atom_gemm_w4a4_multi_merge_q_fits(int64_t M, int64_t N_seg, int nseg, int64_t K_total, int splits) {
  if (M < 1 || nseg < 1 || nseg > 3 || N_seg < 16 || (N_seg % 16) != 0 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0) return 0;
  if (!ATOM_TUNE("ATOM_GEMVQ_MERGE", 1) || M > gemv_tokens(K_total) || !gemvq_merge_fits(M, N_seg * nseg, K_total, splits)) return 0;
  return atom_gemm_w4a4_multi_fits(M, N_seg, nseg, K_total);
}

int atom_gemm_w4a4_multi_merge_q(const void *partials_f32, int splits, const int16_t *reorder_index, float clip, const void *B4, const void *sB,
                                 const void *B8, const void *sB8, void *out0, void *out1, void *out2, unsigned f32_mask,
                                 const void *add0_f16, int64_t M, int64_t N_seg, int nseg, int64_t K_total, int group, int keeper,
                                 void *stream) {
  if (!partials_f32) return ATOM_ERR_INVALID_ARG;
  if (nseg < 1 || nseg > 3 || !out0 || (nseg > 1 && !out1) || (nseg > 2 && !out2)) return ATOM_ERR_INVALID_ARG;
  if (N_seg < 16 || (N_seg % 16) != 0) return ATOM_ERR_SHAPE;
  if ((f32_mask & 1u) && add0_f16) return ATOM_ERR_INVALID_ARG;
  if (!(clip > 0.f)) return ATOM_ERR_INVALID_ARG;
  GemmParams p;
  const int st = fill_params(p, partials_f32, B4, sB, sB, partials_f32, B8, sB, sB8, M, N_seg * nseg, K_total, group, keeper, ATOM_SCALE_LAYOUT_PLAIN);
  if (st != ATOM_OK) return st;
  if (!atom_gemm_w4a4_multi_merge_q_fits(M, N_seg, nseg, K_total, splits)) return ATOM_ERR_SHAPE;
  if (!aligned16(out0) || (out1 && !aligned16(out1)) || (out2 && !aligned16(out2)) || (add0_f16 && !aligned16(add0_f16)) ||
      (reorder_index && !aligned16(reorder_index)))
    return ATOM_ERR_ALIGN;
  p.A4 = nullptr; p.sA = nullptr; p.A8 = nullptr; p.sA8 = nullptr;
  p.seg_out[0] = out0; p.seg_out[1] = out1; p.seg_out[2] = out2;
  p.seg_add = (const half_t *)add0_f16;
  p.seg_n = (int)N_seg;
  p.seg_f32 = f32_mask;
  p.q_op = 5;
  p.q_x = nullptr; p.q_x2 = nullptr; p.q_res = nullptr; p.q_res_out = nullptr;
  p.q_idx = reorder_index;
  p.q_eps = 0.f; p.q_clip = clip;
  p.q_part = (const float *)partials_f32;
  p.q_splits = splits;
#ifdef ATOM_TOOLS   // traced run (tools/r06/gemvq_trace.py)
  if (const char *e = getenv("ATOM_TRACE_PTR")) p.Dsz = reinterpret_cast<half_t *>(strtoull(e, nullptr, 16));
#endif
  return launch_gemvq_multi_q(p, reinterpret_cast<hipStream_t>(stream));
}

size_t atom_gemm_w4a4_o4_workspace_bytes(int64_t M, int64_t N, int64_t K_total) {
  if (M < 1 || N < 128 || (N % 128) != 0 || K_total < 256 || ((K_total - kKeeper) % kGroup) != 0) return 0;
  return skinny_fits(M, N, K_total) ? (size_t)M * (size_t)N * sizeof(float) : 0;
}

int atom_gemm_w4a4_o4_ws(const void *A4, const void *B4, const void *sA, const void *sB, const void *A8, const void *B8,
                         const void *sA8, const void *sB8, void *D_u4, void *D_scale_zero, int64_t M, int64_t N,
                         int64_t K_total, int group, int keeper, int scale_layout, void *workspace, size_t workspace_bytes,
                         void *stream) {
  const size_t need = atom_gemm_w4a4_o4_workspace_bytes(M, N, K_total);
  if (need == 0 || !workspace || workspace_bytes < need || (scale_layout & (ATOM_A_WIDE | ATOM_AB_F6)))
    return atom_gemm_w4a4_o4(A4, B4, sA, sB, A8, B8, sA8, sB8, D_u4, D_scale_zero, M, N, K_total, group, keeper, scale_layout,
                             stream);
  if (!D_u4 || !D_scale_zero) return ATOM_ERR_INVALID_ARG;
  GemmParams p;
  const int st = fill_params(p, A4, B4, sA, sB, A8, B8, sA8, sB8, M, N, K_total, group, keeper, scale_layout);
  if (st != ATOM_OK) return st;
  if (!aligned16(D_u4) || !aligned16(workspace)) return ATOM_ERR_ALIGN;
  p.D4 = (uint8_t *)D_u4;
  p.Dsz = (half_t *)D_scale_zero;
  p.ws = (float *)workspace;
  const int s2 = launch_gemm_skinny_o4(p, reinterpret_cast<hipStream_t>(stream));
  if (s2 != ATOM_ERR_SHAPE) return s2;
  return atom_gemm_w4a4_o4(A4, B4, sA, sB, A8, B8, sA8, sB8, D_u4, D_scale_zero, M, N, K_total, group, keeper, scale_layout,
                           stream);
}

int atom_gemm_w4a4_silu_mul_quant_f6(const void *A_f6, const void *Bgu_f6s, const void *A8, const void *Bgu8, const void *sA8,
                                     const void *sBgu8, int64_t M, int64_t N_inter, int64_t K_total, int group, int keeper,
                                     int quant_mode, float clip, int scale_layout, void *o_outliers, void *o_norms_f6,
                                     void *outlier_scales, void *norm_scales, void *xq, void *stream) {
  if (!A_f6 || !Bgu_f6s || !A8 || !Bgu8 || !sA8 || !sBgu8 || !o_outliers || !o_norms_f6 || !outlier_scales || !norm_scales)
    return ATOM_ERR_INVALID_ARG;
  const int b_pairs = (scale_layout & ATOM_B_SCALE_PAIRS) != 0;
  scale_layout &= ~ATOM_B_SCALE_PAIRS;
  if (scale_layout != ATOM_SCALE_LAYOUT_REF && scale_layout != ATOM_SCALE_LAYOUT_PLAIN) return ATOM_ERR_INVALID_ARG;
  if (quant_mode != ATOM_QUANT_KERNEL && quant_mode != ATOM_QUANT_SIM) return ATOM_ERR_INVALID_ARG;
  if (!(clip > 0.f) || clip > 1.f) return ATOM_ERR_INVALID_ARG;
  if (group != kGroup || keeper != kKeeper) return ATOM_ERR_SHAPE;
  if (M < 1 || M > (1 << 24) || N_inter < 256 || (N_inter % 128) != 0 || N_inter > (1 << 23) || K_total < 256 ||
      ((K_total - kKeeper) % kGroup) != 0 || K_total > (1 << 20))
    return ATOM_ERR_SHAPE;
  if (!aligned16(A_f6) || !aligned16(Bgu_f6s) || !aligned16(A8) || !aligned16(Bgu8) || !aligned16(o_outliers) ||
      !aligned16(o_norms_f6) || (xq && !aligned16(xq)) || (reinterpret_cast<uintptr_t>(sBgu8) & 3u))
    return ATOM_ERR_ALIGN;
  GemmParams p{};
  const int64_t N = 2 * N_inter;
  p.A4 = (const uint8_t *)A_f6; p.B4 = (const uint8_t *)Bgu_f6s;
  p.A8 = (const uint8_t *)A8;   p.B8 = (const uint8_t *)Bgu8;
  p.sA8 = (const half_t *)sA8;  p.sB8 = (const half_t *)sBgu8;
  p.splits = 1;
  p.M = (int)M; p.N = (int)N;
  p.K4h = (int)((K_total - kKeeper) / 2);
  p.G = (int)((K_total - kKeeper) / kGroup);
  p.ref_layout = scale_layout == ATOM_SCALE_LAYOUT_REF;
  p.b_pairs = b_pairs;
  p.f6_rows_a = (M + 255) / 256 * 256;
  p.f6_rows_b = (N + 255) / 256 * 256;                       // == N: N_inter is a multiple of 128
  p.sB32 = reinterpret_cast<const float *>((const uint8_t *)Bgu_f6s + (size_t)p.G * (size_t)p.f6_rows_b * 104);
  p.ldA = (int64_t)atom_scale_size(M, scale_layout);
  p.gu = GateUpOut{(uint8_t *)o_norms_f6, p.f6_rows_a, (int8_t *)o_outliers, (half_t *)outlier_scales, (half_t *)norm_scales,
                   p.ldA, (half_t *)xq, clip, p.ref_layout};
  return launch_gemm_f6_gateup(p, quant_mode == ATOM_QUANT_SIM, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
