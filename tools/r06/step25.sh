#!/bin/bash
# Round 6, step 25: KV splits of 2 tiles (32 splits at context 1024) with the merge kernel taking 32 splits in one batch of loads
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1,2,4 2>&1 | grep "^batch"; }
{
run ATOM_DECODE_MIN_TILES=4
run ATOM_DECODE_MIN_TILES=2
run ATOM_DECODE_MIN_TILES=3
run ATOM_DECODE_MIN_TILES=4
} | tee $O/ab_decode_min_tiles2.txt
