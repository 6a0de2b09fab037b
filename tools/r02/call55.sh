cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python tools/r02/kg_probe.py > gpurun_out/r02/kg_probe4.txt 2>&1
grep -E "^ +[0-9]+x" gpurun_out/r02/kg_probe4.txt | cut -c1-200
timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_gateup.py -q -x 2>&1 | tail -3
