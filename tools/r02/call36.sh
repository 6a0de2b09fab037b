cd $GRAFT_REPO_ROOT
timeout 1500 bash tools/profile_bench.sh 2>&1 | tail -5
ls gpurun_out/prof_summary/
