cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python tools/r02/kg_probe.py > gpurun_out/r02/kg_probe.txt 2>&1
tail -24 gpurun_out/r02/kg_probe.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
