# packed operands through atom_gemm_w4a4_f16_ws: INT8 kernels vs the re-coding route, weight cached or not, by batch size
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for shape in "256 4096 4096" "384 4096 4096" "512 4096 4096" "640 4096 4096" "768 4096 4096" "512 11008 4096" "512 4096 11008" "256 11008 4096" "384 11008 4096"; do
  for minm in 256 100000; do
    for cached in 1 0; do
      if [ $cached = 1 ]; then c="ATOM_WS_CACHED=1"; else c=""; fi
      env $c ATOM_WS=1 ATOM_F6_ROUTE_MIN_M=$minm timeout 60 build/tools/gemm_bench $shape 200 16 2>&1 | grep -E "RESULT|FAIL" | sed "s/^/min_m=$minm cached=$cached: /" | cut -c1-120
    done
  done
done 2>&1 | tee gpurun_out/r03/route_probe.txt
