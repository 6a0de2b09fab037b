for s in "1 4096 4096 50 1" "1 256 256 5 1" "7 320 640 5 7" "16 5120 5120 50 16" "3 13824 5120 50 3" "16 5120 13824 50 16" "2 4096 11008 50 2" "8 4096 4096 50 8"; do build/gemm_bench $s | grep -E "check|RESULT"; done
ATOM_GEMM_VARIANT=2 build/gemm_bench 16 5120 5120 50 0 | grep RESULT
