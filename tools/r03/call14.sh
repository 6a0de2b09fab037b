cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
 for mg in 0 1; do
   echo -n "merged=$mg rep$rep: "; ATOM_F6=1 ATOM_KEEPER_MERGED=$mg timeout 120 build/tools/gemm_bench 4096 4096 4096 300 64 2>&1 | grep -E "RESULT|FAIL|check" | tr '\n' ' ' | cut -c1-230; echo
 done
done
ATOM_KEEPER_MERGED=1 timeout 100 build/tools/trace_f6q 2>&1 | head -6 | cut -c1-330
