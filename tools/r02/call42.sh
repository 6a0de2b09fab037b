cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python tools/r02/kg_probe.py > gpurun_out/r02/kg_probe3.txt 2>&1
tail -23 gpurun_out/r02/kg_probe3.txt | cut -c1-190
timeout 1200 python -m pytest tests/test_gpu_gemm.py -q -x 2>&1 | tail -4
