cd $GRAFT_REPO_ROOT
for abl in 0 1 2 3; do
 for v1 in 1; do
  for M in 64 256; do
   echo -n "ABL=$abl V1=$v1 M=$M: "; ATOM_SKINNY_ABL=$abl ATOM_SKINNY_V1=$v1 timeout 60 build/tools/gemm_bench $M 4096 4096 300 0 2>&1 | grep RESULT | cut -c1-90
  done
 done
done
echo -n "v2 M=64: "; timeout 60 build/tools/gemm_bench 64 4096 4096 300 0 2>&1 | grep RESULT | cut -c1-90
echo -n "v2 M=256: "; timeout 60 build/tools/gemm_bench 256 4096 4096 300 0 2>&1 | grep RESULT | cut -c1-90
