# same-box A/B of build/ab/<names...>: F6 headline at 4096^3 (and a second shape), alternating, 300 launches x 5 each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
names="$@"
for rep in 1 2; do
  for n in $names; do
    for shape in "4096 4096 4096" "8192 8192 8192"; do
      ATOM_F6=1 timeout 120 build/ab/$n/gemm_bench $shape 300 64 2>&1 | grep -E "RESULT|FAIL" | sed "s/^/$n rep$rep: /" | cut -c1-200
    done
  done
done
