for s in "300 320 384 3 300" "1000 1408 640 3 1000" "64 320 1152 3 64"; do build/gemm_bench $s | grep check; done
for v in 0 300; do ATOM_GEMM_VARIANT=$v build/gemm_bench 4096 4096 4096 300 0 | grep RESULT; done
build/gemm_bench 8192 8192 8192 20 0 | grep RESULT
