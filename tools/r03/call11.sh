cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_e2e.py tests/test_gpu_kv.py -m gpu -q -x -k "decode or split or skinny or multi or o4 or decoder or fused or kv or append" 2>&1 | tail -4
timeout 600 python tools/cold_bench.py gemm > gpurun_out/r03/cold_gemm2.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03/cold_gemm2.txt | cut -c1-200
