export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c13; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_gateup.py -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt
cat $O/pytest.txt
