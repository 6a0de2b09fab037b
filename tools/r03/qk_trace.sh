cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 120 build/tools/trace_f6q 1024 4096 4096 qk 2>&1 | tee gpurun_out/r03/trace_f6qk.txt
