cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for abl in 3 5 9 7 11 13 15; do
  ATOM_F6_CFG=$((2100+abl)) ATOM_F6=1 timeout 120 build/tools/gemm_bench 1024 4096 4096 300 0 2>&1 | grep -E "RESULT" | sed "s/^/abl=$abl: /" | cut -c1-120
done 2>&1 | tee gpurun_out/r03/qk_abl2.txt
