for s in "300 320 384 5 300" "4096 4096 4096 30 4096" "1000 11008 4096 20 1000" "777 5120 13824 10 777"; do build/gemm_bench $s | grep -E "check|RESULT"; done
build/gemm_bench 8192 8192 8192 10 0 | grep RESULT
