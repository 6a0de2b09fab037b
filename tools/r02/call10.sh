export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c10; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_block.py tests/test_gpu_gemm.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -30 > $O/pytest.txt
cat $O/pytest.txt
