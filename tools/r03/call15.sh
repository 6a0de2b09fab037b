cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_gateup.py tests/test_gpu_e2e.py tests/test_gpu_pack.py -m gpu -q 2>&1 | tail -25 | cut -c1-300
