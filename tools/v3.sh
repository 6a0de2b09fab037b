for v in 300 301 302 303; do ATOM_GEMM_VARIANT=$v build/gemm_bench 300 320 384 3 300 | grep check; ATOM_GEMM_VARIANT=$v build/gemm_bench 1000 1408 640 3 1000 | grep check; done
for v in 0 300 301 302 303 301; do ATOM_GEMM_VARIANT=$v build/gemm_bench 4096 4096 4096 30 0 | grep RESULT; done
for v in 0 301 302; do ATOM_GEMM_VARIANT=$v build/gemm_bench 8192 8192 8192 10 0 | grep RESULT; ATOM_GEMM_VARIANT=$v build/gemm_bench 2048 11008 4096 20 0 | grep RESULT; done
