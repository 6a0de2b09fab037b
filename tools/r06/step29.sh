#!/bin/bash
# Round 6, step 29: gemvq with the token operand read once (small rings)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/cold_bench.py layer 1,2 2>&1 | grep "^batch"
timeout 600 bash tools/r06/decode_prof.sh step29_b1 1 2>&1 | grep "us avg\|sum of"
