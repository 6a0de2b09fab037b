export ATOM_WS=1
for mm in 16 0; do export ATOM_GEMV_MAXM=$mm; for s in "1 4096 4096 50 0" "2 4096 4096 50 0" "4 4096 4096 50 0" "8 4096 4096 50 0" "16 4096 4096 50 0" "16 5120 5120 50 0" "4 13824 5120 50 0" "16 13824 5120 50 0" "1 13824 5120 50 0" "16 5120 13824 50 0"; do build/gemm_bench $s | grep -E "RESULT" | sed "s/^/maxm=$mm /"; done; done
