#!/bin/bash
# Round 6, step 32: two-level KV-split merge at small batches (4 waves per workgroup -> one partial state): tests, then the layer with
# the merge launch over 4 partial states and with the merge inside o_proj's launch
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -3
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1,2,4 2>&1 | grep "^batch"; }
{
run ATOM_DECODE_INNER=0
run ATOM_DECODE_INNER=1
run ATOM_DECODE_INNER=1 ATOM_MERGE_IN_O_PROJ=1
run ATOM_DECODE_INNER=0
run ATOM_DECODE_INNER=1 ATOM_MERGE_IN_O_PROJ=1
} | tee $O/ab_decode_inner.txt
