# round-end pass: full GPU tests, bench line, PMC / stats profile of the bench command, 7B block parity output
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r03/pytest_gpu_full.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_line.json 2> gpurun_out/r03/bench_err.txt
tail -c 600 gpurun_out/r03/bench_line.json
timeout 1500 bash tools/profile_bench.sh 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_block.py -m gpu -q -x -s -k "7b_block" > gpurun_out/r03/block7b.txt 2>&1
grep -E "passed|failed" gpurun_out/r03/block7b.txt
