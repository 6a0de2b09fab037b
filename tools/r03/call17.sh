cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_gateup.py -m gpu -q -x 2>&1 | tail -3
bash tools/r03/ab.sh k1 k2
