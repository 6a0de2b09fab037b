#!/bin/bash
# Round 6, step 22: the output addend (residual stream) requested with the weights in the dot-product and decode-batch kernels
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/cold_bench.py layer 1,4,8,16 2>&1 | grep "^batch" | tee $O/layer_step22.txt
timeout 600 bash tools/r06/decode_prof.sh step22_b1 1 2>&1 | grep "us avg\|sum of"
timeout 600 bash tools/r06/decode_prof.sh step22_b16 16 2>&1 | grep "skinny\|sum of"
