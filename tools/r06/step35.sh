#!/bin/bash
# Round 6, step 35: kernel arguments (incl. the implicit gridDim) in one batch in the one-token GEMV and the dot-product kernel
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -2
{
timeout 300 python tools/cold_bench.py gemm 1 2>&1 | grep -v amdgpu | head -12
timeout 300 python tools/cold_bench.py layer 1,2 2>&1 | grep "^batch"
} | tee $O/step35.txt
