cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/prof2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o block -- python $R/bench.py --workload block --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pb.log 2>&1
cp /tmp/pb/block_kernel_stats.csv $R/gpurun_out/r02/prof2/block_kernel_stats.csv 2>/dev/null
head -25 /tmp/pb/block_kernel_stats.csv | cut -c1-200
for M in 256 512 1024; do
  ATOM_F6=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm$M -o m -- $R/build/tools/gemm_bench $M 4096 4096 300 0 > /dev/null 2>&1
  echo "## M=$M"; head -3 /tmp/pm$M/m_kernel_stats.csv | cut -c1-220
  head -3 /tmp/pm$M/m_kernel_stats.csv > $R/gpurun_out/r02/prof2/midm_${M}_kernel_stats.csv
done
