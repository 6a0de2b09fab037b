cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python tools/r02/kg_probe.py > gpurun_out/r02/kg_probe2.txt 2>&1
tail -24 gpurun_out/r02/kg_probe2.txt
