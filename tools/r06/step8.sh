#!/bin/bash
# Round 6, step 8: KV splits as waves of 12-wave workgroups (pairs x splits), merge in LDS: tests, the layer by batch, A/B against two launches.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_step8.txt
{
echo "== splits as waves of 12-wave workgroups (default)"
timeout 600 python tools/cold_bench.py layer 1,8,16,32,64 2>&1 | grep "^batch"
echo "== split kernel + merge launch (ATOM_DECODE_WGM_PAIRS=100000000, tools build)"
ATOM_LIB=$PWD/build/tools/libatom_hip.so ATOM_DECODE_WGM_PAIRS=100000000 timeout 600 python tools/cold_bench.py layer 8,16,32 2>&1 | grep "^batch"
} | tee $O/ab_decode_wgm.txt
timeout 600 bash tools/r06/decode_prof.sh step8_b16 16 2>&1 | tail -14
