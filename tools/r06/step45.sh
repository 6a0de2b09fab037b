#!/bin/bash
# Round 6, step 45: with the merge inside o_proj's launch, 8 split waves per workgroup (2 partial states per head at context 1024) against 4 (4 states)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1,2 2>&1 | grep "^batch"; }
{
for r in 1 2; do
run ATOM_LIB=$PWD/build/tools/libatom_hip.so ATOM_MERGE_IN_O_PROJ=1
run ATOM_LIB=$PWD/build/ab/inner8/libatom_hip.so ATOM_MERGE_IN_O_PROJ=1
done
} | tee $O/ab_merge_in_o_proj_inner8.txt
