cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_gemm.py -q -x -k "random_shapes or two_k_group or properties or f6" 2>&1 | tail -8
