for v in 0 1128 1256 0 1128 1256; do ATOM_GEMM_VARIANT=$v build/gemm_bench 4096 4096 4096 30 0 | grep RESULT; done
