cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02/bench_line.json 2> gpurun_out/r02/bench_err.txt; tail -c 3000 gpurun_out/r02/bench_line.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
