#!/bin/bash
# Round 6, step 36: kernel arguments in one batch in the stand-alone quantiser kernels
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_quant.py tests/test_gpu_e2e.py tests/test_gpu_block.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tools/cold_bench.py layer 1,2,4,16 2>&1 | grep "^batch" | tee $O/step36.txt
timeout 600 bash tools/r06/decode_prof.sh step36_b16 16 2>&1 | grep "quant2\|sum of"
