cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 600 python bench.py --workload block --steps 3 --warmup 1 > gpurun_out/r02/bench_block.json 2> gpurun_out/r02/bench_block_err.txt; tail -c 1500 gpurun_out/r02/bench_block.json
timeout 600 python tools/block_bench.py 2>&1 | tail -14
