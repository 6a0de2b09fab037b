ATOM_GEMM_VARIANT=1064 build/gemm_bench 300 320 384 3 300 | grep check
for v in 0 1064 0 1064; do ATOM_GEMM_VARIANT=$v build/gemm_bench 4096 4096 4096 30 0 | grep RESULT; done
ATOM_GEMM_VARIANT=1064 build/gemm_bench 8192 8192 8192 10 0 | grep RESULT
