cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x 2>&1 | tail -3
bash tools/r03/ab.sh old new new2
