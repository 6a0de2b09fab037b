cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 1500 bash tools/profile_bench.sh > gpurun_out/r03/profile_bench.log 2>&1
tail -3 gpurun_out/r03/profile_bench.log
ls gpurun_out/prof_summary
