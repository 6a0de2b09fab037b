for v in 304 305 306; do for s in "64 320 384 3 64" "17 1408 640 3 17" "100 256 1152 3 100"; do ATOM_GEMM_VARIANT=$v build/gemm_bench $s | grep check; done; done
for s in "16 5120 5120 50 0" "64 5120 5120 50 0" "64 13824 5120 30 0" "64 5120 13824 30 0" "32 4096 4096 50 0" "128 4096 4096 50 0"; do for v in 0 304 305 306 302; do ATOM_GEMM_VARIANT=$v build/gemm_bench $s | grep RESULT; done; done
