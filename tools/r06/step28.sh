#!/bin/bash
# Round 6, step 28: the KV split count at batches 32 / 64 (policy: 3, as waves of 12-wave workgroups) against 1, 2, 4, 6
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 32,64 2>&1 | grep "^batch"; }
{
run ATOM_X=0
run ATOM_DECODE_SPLITS=1
run ATOM_DECODE_SPLITS=2
run ATOM_DECODE_SPLITS=4
run ATOM_DECODE_SPLITS=6
run ATOM_X=0
} | tee $O/ab_decode_splits_large_batch.txt
