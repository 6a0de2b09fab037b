#!/bin/bash
# Round 6, step 11: small batches -- the KV splits of a head as the 12 waves of ONE workgroup (32 heads = 32 CUs, merge in LDS, no merge
# launch) against 16 single-wave workgroups per head + merge launch.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer ${B:-1,2,4} 2>&1 | grep "^batch"; }
{
run ATOM_DECODE_DEFAULT=1
run ATOM_DECODE_WGM_PAIRS=1 ATOM_DECODE_SPLITS=12
run ATOM_DECODE_WGM_PAIRS=1 ATOM_DECODE_SPLITS=6
run ATOM_DECODE_DEFAULT=1
} | tee $O/ab_decode_wgm_small.txt
