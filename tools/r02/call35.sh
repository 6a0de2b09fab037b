cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 300 python tools/decode_bench.py > gpurun_out/r02/decode_bench.txt 2>&1; grep DRESULT gpurun_out/r02/decode_bench.txt
timeout 300 python tools/decode_graph_bench.py > gpurun_out/r02/decode_graph_bench.txt 2>&1; tail -12 gpurun_out/r02/decode_graph_bench.txt
timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_e2e.py -q 2>&1 | tail -4
