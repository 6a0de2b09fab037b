cd $GRAFT_REPO_ROOT
for i in 1 2; do
for nt in 0 1; do echo "NT=$nt"; ATOM_NT=$nt ATOM_F6=1 timeout 120 build/tools/gemm_bench 4096 4096 4096 300 64 | grep -E "RESULT|bad"; done
done
