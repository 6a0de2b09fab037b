for v in 204 1031; do for K in 256 1152 2176 4096; do ATOM_GEMM_VARIANT=$v build/gemm_bench 4096 4096 $K 30 0 | grep RESULT; done; done
