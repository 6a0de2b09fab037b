cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 300 build/tools/bank_probe > gpurun_out/r03/bank_probe.txt 2>&1
cat gpurun_out/r03/bank_probe.txt
