for v in 0 203 204; do ATOM_GEMM_VARIANT=$v build/gemm_bench 512 512 640 5 512 | grep -E "check|RESULT"; done
for v in 203 204; do ATOM_GEMM_VARIANT=$v build/gemm_bench 300 320 384 5 300 | grep -E "check"; ATOM_GEMM_VARIANT=$v build/gemm_bench 4096 4096 4096 30 4096 | grep -E "check|RESULT"; ATOM_GEMM_VARIANT=$v build/gemm_bench 2048 11008 4096 20 0 | grep RESULT;  ATOM_GEMM_VARIANT=$v build/gemm_bench 8192 8192 8192 10 0 | grep RESULT; done
ATOM_GEMM_VARIANT=0 build/gemm_bench 8192 8192 8192 10 0 | grep RESULT
