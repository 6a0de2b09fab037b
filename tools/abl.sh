for v in 0 101 102 103 104 108 110 111 115; do ATOM_GEMM_VARIANT=$v build/gemm_bench 4096 4096 4096 30 0 | grep RESULT; done
