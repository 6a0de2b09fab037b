#!/bin/bash
# Round 6, step 7: the decode attention's tile ring (DP tiles requested per wave) x KV splits as separate workgroups + merge launch / as the
# waves of one workgroup (WGM): the Llama-7B decode layer by batch (tools build; every variant is bit-identical for a given split count).
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer ${B:-1,16,64} 2>&1 | grep "^batch"; }
{
run ATOM_DECODE_WGM_PAIRS=100000000 ATOM_DECODE_DP=3
run ATOM_DECODE_WGM_PAIRS=100000000 ATOM_DECODE_DP=4
run ATOM_DECODE_WGM_PAIRS=100000000 ATOM_DECODE_DP=6
B=16,64
run ATOM_DECODE_SPLITS=4 ATOM_DECODE_WGM_PAIRS=100000000 ATOM_DECODE_DP=4
run ATOM_DECODE_SPLITS=4 ATOM_DECODE_DP=4
run ATOM_DECODE_SPLITS=4 ATOM_DECODE_DP=6
run ATOM_DECODE_SPLITS=8 ATOM_DECODE_DP=4
run ATOM_DECODE_SPLITS=8 ATOM_DECODE_DP=6
run ATOM_DECODE_SPLITS=2 ATOM_DECODE_DP=6
} | tee $O/ab_decode_ring.txt
