cd $GRAFT_REPO_ROOT
ATOM_WS=1 timeout 120 build/tools/gemm_bench 2048 13824 5120 200 16 | grep RESULT
ATOM_WS=1 timeout 120 build/tools/gemm_bench 4096 4096 4096 200 16 | grep RESULT
