for v in 0 203 1032 1001 1033 1003 1035; do ATOM_GEMM_VARIANT=$v build/gemm_bench 4096 4096 4096 30 0 | grep RESULT; done
