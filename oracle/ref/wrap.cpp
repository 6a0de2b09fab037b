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

extern "C" {

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
