// oracle/_ref/libatom_ref.so: C entry points around the reference's own CPU golden functions (the ones its CUDA unit tests
// check the kernels against), compiled from the reference sources where they lie (oracle/ref/extract.py -> oracle/_ref/*.inc).
// Test infrastructure only: pins the "kernel-flavoured" quantisation arithmetic (FP32, max/7, round(x / scale), clamp) of
//   Reorder   e2e/punica-atom/punica/ops/csrc/Reorder/test_Reorder.cu:41-112
//   Activate  e2e/punica-atom/punica/ops/csrc/Activate/test_activate.cu:41-112
//   RMSNorm   e2e/punica-atom/punica/ops/csrc/Norm/test_RMSNorm.cu:122-195
// Each golden brings its own copies of PackInt4 / clamp / scale_index / SCALE_SIZE_A, so each lives in its own namespace.
#include "shim.h"

namespace ref_reorder {
#include "../_ref/reorder.inc"
#undef SCALE_SIZE_A
}
namespace ref_activate {
#include "../_ref/activate.inc"
#undef SCALE_SIZE_A
}
namespace ref_rmsnorm {
#include "../_ref/rmsnorm.inc"
#undef SCALE_SIZE_A
}

// The reference's CPU restatements of the INT4 paged KV cache (kernels/src/flashinfer/cpu_reference.h:115-234: append_paged_kv_cache,
// single_quantize_mha -> single_mha, apply_llama_rope) -- what its own tests check append_kv_i4 / batch_decode_i4 against
// (kernels/src/flashinfer/bench_batch_decode.cu) -- in the namespaces they are written for.
namespace ref_kv {
namespace flashinfer {
namespace quant {
#include "../_ref/kvquant.inc"
}
#include "../_ref/kvtypes.inc"
}  // namespace flashinfer
namespace cpu_reference {
using namespace flashinfer;
#include "../_ref/kvcpu.inc"
}  // namespace cpu_reference
}  // namespace ref_kv

// The u4-output GEMM's epilogue (DenseLayerGEMM_i4_o4.cu:704-788).  Only local_max_min (:72-80, __host__ __device__) exists as
// host-callable reference code; the rest is the tail of the __global__ kernel.  o4_tile_row() applies that tail's scalar lines, in
// their order and with their expressions, to one row of one 128-column output block, with the two threads of the row (loc_idx 0 /
// 1, 64 elements each, :712-720) run one after the other and the shfl.bfly exchange (:737-747) as a plain swap -- around the
// reference's own local_max_min, PackInt4, mymax and mymin.
namespace ref_o4 {
using std::abs;                                            // device code resolves abs((float)x) to the float overload
#include "../_ref/o4.inc"
constexpr int GROUP_SIZE = 128, elements_per_thread = 64, loadIters = elements_per_thread * sizeof(float) / 16;   // :708, :719-720
inline void o4_tile_row(const float *row /* 128 FP32 sums */, uint8_t *D_row /* 64 bytes */, half2 *scale_zero) {
  float4 elements_float4[2][loadIters];
  float local_max[2], local_min[2];
  for (int loc_idx = 0; loc_idx < 2; ++loc_idx) {
    for (int i = 0; i < loadIters; ++i)                    // :722-727
      elements_float4[loc_idx][i] = *(reinterpret_cast<const float4 *>(row + loc_idx * GROUP_SIZE / 2) + i);
    local_max[loc_idx] = __half2float(half(-INFINITY));    // :729  0xFC00
    local_min[loc_idx] = __half2float(half(INFINITY));     // :730  0x7C00
    for (int i = 0; i < loadIters; ++i)                    // :732-735
      local_max_min<float, float4>(elements_float4[loc_idx] + i, local_max[loc_idx], local_min[loc_idx]);
  }
  const float pmax[2] = {local_max[1], local_max[0]}, pmin[2] = {local_min[1], local_min[0]};   // :737-747 (bfly, lane ^ 1)
  for (int loc_idx = 0; loc_idx < 2; ++loc_idx) {
    float temp = pmax[loc_idx];
    local_max[loc_idx] = mymax(local_max[loc_idx], temp);
    temp = pmin[loc_idx];
    local_min[loc_idx] = mymin(local_min[loc_idx], temp);
    float scale = (local_max[loc_idx] - local_min[loc_idx]) / 15.f;   // :749-751
    float zero = -local_min[loc_idx];
    float r_scale = 1.f / scale;
    if (loc_idx == 0) *scale_zero = __floats2half2_rn(scale, zero);   // :753-759
    PackInt4 quantized[elements_per_thread / 2];
    float *elements_float = reinterpret_cast<float *>(elements_float4[loc_idx]);
    for (int i = 0; i < elements_per_thread; i += 2) {                // :762-771
      int8_t result_0, result_1;
      result_0 = (int8_t)(round((elements_float[i] + zero) * r_scale));
      result_1 = (int8_t)(round((elements_float[i + 1] + zero) * r_scale));
      quantized[i / 2].low = result_0 & 0xf;
      quantized[i / 2].high = result_1 & 0xf;
    }
    std::memcpy(D_row + loc_idx * GROUP_SIZE / 2 / 2, quantized, sizeof(quantized));   // :774-786
  }
}
}  // namespace ref_o4

extern "C" {

// c32 float [M, N] (the GEMM's FP32 sums) -> D u8 [M, N/2], output_scale half2 (scale, zero) [M, N/128]
// (D + row * N/2 + bi * 64 + loc_idx * 32, output_scale[row * N/128 + bi]: DenseLayerGEMM_i4_o4.cu:758, :781-783)
void ref_cpu_o4_epilogue(const float *c32, int M, int N, uint8_t *D, void *output_scale) {
  for (int row = 0; row < M; ++row)
    for (int bi = 0; bi < N / 128; ++bi)
      ref_o4::o4_tile_row(c32 + (size_t)row * N + bi * 128, D + (size_t)row * N / 2 + bi * 64,
                          reinterpret_cast<half2 *>(output_scale) + (size_t)row * N / 128 + bi);
}

// Append: cache u8 [pages, L, 2, N, P, D/2] / param half2 [pages, L, 2, N, P] (punica/utils/kvcache.py:17-26); k, v u8 [T, N, D/2] and
// k_param, v_param half2 [T, N] hold the new tokens of all sequences back to back (append_indptr int32 [B + 1])
void ref_cpu_append_paged_kv_i4(uint8_t *data, void *param, int32_t *indptr, int32_t *indices, int32_t *last_page_offset,
                                int num_layers, int layer, int num_heads, int page_size, int head_dim, int batch,
                                const uint8_t *k, const uint8_t *v, const void *k_param, const void *v_param,
                                const int32_t *append_indptr) {
  using namespace ref_kv;
  using s4 = flashinfer::quant::__precision__s4;
  flashinfer::paged_kv_t<s4, int32_t> pg(num_layers, layer, num_heads, page_size, head_dim, batch, reinterpret_cast<s4 *>(data),
                                         reinterpret_cast<half2 *>(param), indptr, indices, last_page_offset);
  std::vector<std::vector<uint8_t>> ks(batch), vs(batch);
  std::vector<std::vector<half2>> kp(batch), vp(batch);
  const size_t tok = (size_t)num_heads * head_dim / 2;
  for (int b = 0; b < batch; ++b) {
    const size_t t0 = append_indptr[b], n = append_indptr[b + 1] - append_indptr[b];
    ks[b].assign(k + t0 * tok, k + (t0 + n) * tok);
    vs[b].assign(v + t0 * tok, v + (t0 + n) * tok);
    kp[b].assign((const half2 *)k_param + t0 * num_heads, (const half2 *)k_param + (t0 + n) * num_heads);
    vp[b].assign((const half2 *)v_param + t0 * num_heads, (const half2 *)v_param + (t0 + n) * num_heads);
  }
  std::vector<int32_t> ai(append_indptr, append_indptr + batch + 1);
  cpu_reference::append_paged_kv_cache<s4, int32_t>(pg, ks, vs, kp, vp, ai);
}

// One query token against kv_len quantised tokens, layout NHD: q half [N, D]; k, v u8 [kv_len, N, D/2]; params half2 [kv_len, N];
// llama RoPE on q (at position kv_len - 1) and on every key at its own position; out float [N, D]
void ref_cpu_single_decode_i4(const void *q, const uint8_t *k, const uint8_t *v, const void *k_param, const void *v_param, int kv_len,
                              int num_heads, int head_dim, float rope_scale, float rope_theta, float *out) {
  using namespace ref_kv;
  const size_t nq = (size_t)num_heads * head_dim, nk = (size_t)kv_len * num_heads * head_dim / 2, np = (size_t)kv_len * num_heads;
  std::vector<half> qv((const half *)q, (const half *)q + nq);
  std::vector<uint8_t> kv(k, k + nk), vv(v, v + nk);
  std::vector<half2> kp((const half2 *)k_param, (const half2 *)k_param + np), vp((const half2 *)v_param, (const half2 *)v_param + np);
  std::vector<float> o = cpu_reference::single_quantize_mha<half, uint8_t, float>(
      qv, kv, vv, kp, vp, 1, kv_len, num_heads, head_dim, false, flashinfer::QKVLayout::kNHD, flashinfer::RotaryMode::kLlama, rope_scale,
      rope_theta);
  for (size_t i = 0; i < nq; ++i) out[i] = o[i];
}

// x fp16 [seq_len, hidden]; reorder_index int16 [hidden]; o_outliers int8 [seq_len, group]; o_norms packed int4
// [seq_len, (hidden - group) / 2]; scales in the reference's replicated layout (outlier: scale_size(seq_len) halves,
// norm: (hidden / group - 1) * scale_size(seq_len)); the buffers must be zero-filled by the caller (gaps are not written)
void ref_cpu_reorder_fp16_i4(void *x, int group_size, int hidden_dim, int seq_len, int16_t *reorder_index, int8_t *o_outliers,
                             int8_t *o_norms, void *outlier_scales, void *norm_scales) {
  ref_reorder::run_cpu_reorder_fp16_i4((half *)x, group_size, hidden_dim, seq_len, reorder_index, o_outliers, o_norms,
                                       (half *)outlier_scales, (half *)norm_scales);
}

void ref_cpu_activate_fp16_i4(void *a, void *b, int group_size, int hidden_dim, int seq_len, int8_t *o_outliers, int8_t *o_norms,
                              void *outlier_scales, void *norm_scales) {
  ref_activate::run_cpu_activate_fp16_i4((half *)a, (half *)b, group_size, hidden_dim, seq_len, o_outliers, o_norms,
                                         (half *)outlier_scales, (half *)norm_scales);
}

void ref_cpu_rmsnorm_fp16_i4(void *x, void *weight, float eps, int group_size, int hidden_dim, int seq_len, int16_t *reorder_index,
                             int8_t *o_outliers, int8_t *o_norms, void *outlier_scales, void *norm_scales) {
  ref_rmsnorm::run_cpu_rmsnorm_fp16_i4((half *)x, (half *)weight, eps, group_size, hidden_dim, seq_len, reorder_index, o_outliers,
                                       o_norms, (half *)outlier_scales, (half *)norm_scales);
}

int ref_scale_index(int row) { return ref_reorder::scale_index(row); }

}  // extern "C"
