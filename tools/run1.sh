mkdir -p gpurun_out/r1
export TMPDIR=/tmp
( build/gemm_bench 256 256 640 5 256; build/gemm_bench 4096 4096 4096 50 4096; build/gemm_bench 1 4096 4096 50 1; build/gemm_bench 2048 11008 4096 20 0 ) > gpurun_out/r1/gemm_bench.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r1/smoke.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -x -k "not full_size and not properties" > gpurun_out/r1/pytest_small.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -k "full_size or properties" > gpurun_out/r1/pytest_full.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/r1/bench.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r1/prof -o gemm -- $GRAFT_REPO_ROOT/build/gemm_bench 4096 4096 4096 50 0 > $GRAFT_REPO_ROOT/gpurun_out/r1/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -5 gpurun_out/r1/gemm_bench.log gpurun_out/r1/smoke.log gpurun_out/r1/pytest_small.log gpurun_out/r1/pytest_full.log gpurun_out/r1/bench.log
