cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 200 build/tools/trace_f6q > gpurun_out/r03/trace_f6q.txt 2>&1
cat gpurun_out/r03/trace_f6q.txt | cut -c1-400
