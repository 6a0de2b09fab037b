export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c15; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest.txt
python tools/block_bench.py > $O/block.txt 2>&1
cat $O/pytest.txt $O/block.txt
