# per-kernel times of the decode layer under graph replay (cold weights): rocprofv3 kernel trace of tools/cold_bench.py layer
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03/layerprof
for b in 1 16; do
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/lp$b -o lp -- python $R/tools/cold_bench.py layer $b > /tmp/lp$b.log 2>&1
  tail -3 /tmp/lp$b.log
  f=$(find /tmp/lp$b -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/r03/layerprof/b${b}_kernel_stats.csv
  head -25 $f | cut -c1-200
done
