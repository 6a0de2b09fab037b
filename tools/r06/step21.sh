#!/bin/bash
# Round 6, step 21: the merge kernel with 16 splits in one batch of loads
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/cold_bench.py layer 1,2,4 2>&1 | grep "^batch" | tee $O/layer_step21.txt
timeout 600 bash tools/r06/decode_prof.sh step21_b1 1 2>&1 | grep "merge\|batch_decode\|sum of"
