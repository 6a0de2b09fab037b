#!/bin/bash
# Round 6, step 6: decode attention with the splits as the waves of one workgroup (merge in LDS) and the packed q rotation:
# KV / e2e tests, the layer by batch, the per-kernel split at batch 16.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_step6.txt
timeout 600 python tools/cold_bench.py layer 1,4,8,16,64 2>&1 | tee $O/layer_step6.txt
ATOM_LIB=$PWD/build/tools/libatom_hip.so ATOM_DECODE_WGM_PAIRS=100000000 timeout 600 python tools/cold_bench.py layer 8,16,64 2>&1 | sed 's/^/[two launches] /' | tee -a $O/layer_step6.txt
timeout 600 bash tools/r06/decode_prof.sh step6_b16 16 2>&1 | tail -14
