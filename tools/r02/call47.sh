cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_gpu_kv.py -q 2>&1 | tail -4
timeout 600 python tools/decode_step_bench.py > gpurun_out/r02/decode_step.txt 2>&1; tail -12 gpurun_out/r02/decode_step.txt | cut -c1-200
