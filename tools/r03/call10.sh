cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kv.py -m gpu -q -x 2>&1 | tail -15
timeout 600 python tools/cold_bench.py layer > gpurun_out/r03/cold_layer2.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03/cold_layer2.txt | cut -c1-200
