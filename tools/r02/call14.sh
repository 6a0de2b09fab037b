export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c14; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_gateup.py tests/test_gpu_e2e.py tests/test_gpu_block.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest.txt
python tools/r02/gateup_bench.py > $O/gateup.txt 2>&1
python bench.py --workload block --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-900 > $O/block.json
cat $O/pytest.txt $O/gateup.txt $O/block.json
