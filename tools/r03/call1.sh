cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench0.json 2> gpurun_out/r03/bench0.err; tail -c 1500 gpurun_out/r03/bench0.json
