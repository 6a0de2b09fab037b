cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -x -k "headline_kernel_vs_int8" 2>&1 | tail -5
