#!/usr/bin/env python3
"""Build step of oracle/_ref (test infrastructure): cut the reference's OWN CPU golden functions out of its CUDA test files,
where they lie under /root/reference, into generated include files under oracle/_ref/ (git-ignored; reference sources are
never committed to this repository).  The .cu files as a whole cannot be compiled here (nvcc, <<<>>> launches,
cooperative_groups); the CPU goldens inside them are plain C++ once `half` is a host type (oracle/ref/shim.h).

    python oracle/ref/extract.py /root/reference oracle/_ref

A definition is located by its first line (regex) and ends at the first line that closes it at column 0 (`}` or `};`);
#define lines are taken alone.  The extraction fails loudly if a pattern is not found exactly once.
"""
import os
import re
import sys

CSRC = "e2e/punica-atom/punica/ops/csrc"
KINC = "kernels/include/flashinfer"
KSRC = "kernels/src/flashinfer"
# generated file -> [(reference file, first-line regex, kind)]
SPEC = {
    "reorder.inc": [
        ("Reorder/Reorder.cu", r"^struct PackInt4 \{", "block"),
        ("Reorder/Reorder.cu", r"^HOST_DEVICE int clamp\(", "line"),
        ("Reorder/Reorder.cu", r"^template <typename T> HOST_DEVICE T abs\(", "line"),
        ("Reorder/Reorder.cu", r"^HOST_DEVICE int scale_index\(", "block"),
        ("Reorder/Reorder.cu", r"^#define SCALE_SIZE_A\(", "line"),
        ("Reorder/test_Reorder.cu", r"^void run_cpu_reorder_fp16_i4\(", "block"),
    ],
    "activate.inc": [
        ("Activate/Activate.cu", r"^struct PackInt4 \{", "block"),
        ("Activate/Activate.cu", r"^HOST_DEVICE int clamp\(", "line"),
        ("Activate/Activate.cu", r"^template <typename T> HOST_DEVICE T abs\(", "line"),
        ("Activate/Activate.cu", r"^HOST_DEVICE float silu\(", "line"),
        ("Activate/Activate.cu", r"^HOST_DEVICE int scale_index\(", "block"),
        ("Activate/Activate.cu", r"^#define SCALE_SIZE_A\(", "line"),
        ("Activate/test_activate.cu", r"^void run_cpu_activate_fp16_i4\(", "block"),
    ],
    "rmsnorm.inc": [
        ("Norm/RMSNorm.cu", r"^struct PackInt4 \{", "block"),
        ("Norm/RMSNorm.cu", r"^HOST_DEVICE int clamp\(", "line"),
        ("Norm/RMSNorm.cu", r"^template <typename T> HOST_DEVICE T abs\(", "line"),
        ("Norm/RMSNorm.cu", r"^HOST_DEVICE int scale_index\(", "block"),
        ("Norm/RMSNorm.cu", r"^#define SCALE_SIZE_A\(", "line"),
        ("Norm/test_RMSNorm.cu", r"^void run_cpu_rmsnorm_fp16_i4\(", "block"),
    ],
    # the u4-output GEMM's epilogue helpers (DenseLayerGEMM_i4_o4.cu:63-80): PackInt4, mymax / mymin and local_max_min, which is
    # __host__ __device__ -- the one piece of that epilogue that exists as host-callable reference code (the rest, :729-771, is
    # the tail of a __global__ kernel and is applied line by line in oracle/ref/wrap.cpp around this function)
    "o4.inc": [
        ("GEMM/DenseLayerGEMM_i4_o4.cu", r"^struct PackInt4\{", "block"),
        ("GEMM/DenseLayerGEMM_i4_o4.cu", r"^#define mymax\(", "line"),
        ("GEMM/DenseLayerGEMM_i4_o4.cu", r"^#define mymin\(", "line"),
        ("GEMM/DenseLayerGEMM_i4_o4.cu", r"^__forceinline__ __host__ __device__ void local_max_min\(", "block"),
    ],
    # the reference's CPU implementations of the INT4 paged KV cache append and of quantised attention (its own test oracle,
    # kernels/src/flashinfer/cpu_reference.h) with the few definitions they use (paths relative to the reference root), one file
    # per namespace they live in
    "kvquant.inc": [                                         # namespace flashinfer::quant
        ("/" + KINC + "/quantization.cuh", r"^\s+struct __precision__s4\{\};", "line"),
        ("/" + KINC + "/quantization.cuh", r"^\s+FLASHINFER_INLINE constexpr float size_of_type\(\)\{", "iblock"),
        ("/" + KINC + "/quantization.cuh", r"^\s+FLASHINFER_INLINE T\* get_ptr\(T\* ptr, const size_t offset\)\{", "iblock"),
    ],
    "kvtypes.inc": [                                         # namespace flashinfer
        ("/" + KINC + "/layout.cuh", r"^enum class QKVLayout \{", "block"),
        ("/" + KINC + "/layout.cuh", r"^__host__ __device__ __forceinline__ size_t get_elem_offset_impl\(", "block"),
        ("/" + KINC + "/layout.cuh", r"^struct tensor_info_t \{", "block"),
        ("/" + KINC + "/rope.cuh", r"^enum class RotaryMode \{", "block"),
        ("/" + KINC + "/utils.cuh", r"^#define SWITCH_LAYOUT\(", "macro"),
        ("/" + KINC + "/page.cuh", r"^struct paged_kv_t \{", "block"),
    ],
    "kvcpu.inc": [                                           # namespace cpu_reference
        ("/" + KSRC + "/cpu_reference.h", r"^inline std::vector<float> apply_llama_rope\(", "block"),
        ("/" + KSRC + "/cpu_reference.h", r"^std::vector<dtype_out> single_mha\(", "block"),
        ("/" + KSRC + "/cpu_reference.h", r"^void append_paged_kv_cache\(", "block"),
        ("/" + KSRC + "/cpu_reference.h", r"^std::vector<OType> single_quantize_mha\(", "block"),
    ],
}


def cut(lines, pattern, kind, where):
    hits = [i for i, l in enumerate(lines) if re.search(pattern, l)]
    if len(hits) != 1:
        raise SystemExit(f"extract.py: {pattern!r} found {len(hits)} times in {where}")
    i = hits[0]
    if kind == "line":
        return i, i + 1
    if kind == "iblock":                                     # an indented definition: ends at `}` of the same indentation
        ind = len(lines[i]) - len(lines[i].lstrip())
        j = i
        while not re.match(r"^\s{%d}\};?\s*$" % ind, lines[j]):
            j += 1
        while i > 0 and re.match(r"^\s*template\s*<", lines[i - 1]):
            i -= 1
        return i, j + 1
    if kind == "macro":                                      # a #define with line continuations
        j = i
        while lines[j].rstrip().endswith("\\"):
            j += 1
        return i, j + 1
    while i > 0 and re.match(r"^template\s*<", lines[i - 1]):   # a template header belongs to the definition
        i -= 1
    j = hits[0]
    while not re.match(r"^\};?\s*$", lines[j]):
        j += 1
        if j >= len(lines):
            raise SystemExit(f"extract.py: no closing brace for {pattern!r} in {where}")
    return i, j + 1


def main(ref_root, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    for name, items in SPEC.items():
        parts = [f"// GENERATED by oracle/ref/extract.py from {ref_root}/{CSRC} -- do not commit\n"]
        for rel, pattern, kind in items:
            path = ref_root + rel if rel.startswith("/") else os.path.join(ref_root, CSRC, rel)
            lines = open(path).read().splitlines(keepends=True)
            a, b = cut(lines, pattern, kind, rel)
            parts.append(f"// ---- {rel}:{a + 1}-{b}\n")
            parts.extend(lines[a:b])
            parts.append("\n")
        # SCALE_SIZE_A is a macro: un-define it at the end of each namespace's include so the three copies do not clash
        open(os.path.join(out_dir, name), "w").write("".join(parts))
    print("extracted", ", ".join(sorted(SPEC)), "->", out_dir)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
