#!/bin/bash
# Round 6, step 23: the decode attention's tile pipeline rebuilt -- page entries by one vector load + v_readlane, three buffers in rotation,
# unconditional refills (exact s_waitcnt vmcnt counts in the loop)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_e2e.py tests/test_gpu_block.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/cold_bench.py layer 1,2,4,8,16,32,64 2>&1 | grep "^batch" | tee $O/layer_step23.txt
timeout 600 bash tools/r06/decode_prof.sh step23_b16 16 2>&1 | grep "batch_decode\|sum of"
timeout 600 bash tools/r06/decode_prof.sh step23_b1 1 2>&1 | grep "batch_decode\|merge\|sum of"
timeout 300 python tools/decode_bench.py 2>&1 | grep DRESULT
