#!/bin/bash
# Round 6, step 58: the streamers' ring requested in FRONT of the workgroup barrier (ahead of the quantiser's requests), now that the ring is
# half a share: gate / up only (1), the other roles launches (2), all (3); head = the barrier where it was
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1 2>&1 | grep "^batch"; }
{
for r in 1 2; do
run ATOM_LIB=$PWD/build/ab/head/libatom_hip.so
run ATOM_LIB=$PWD/build/tools/libatom_hip.so ATOM_GEMVQ_EARLY_STREAM=0
run ATOM_LIB=$PWD/build/tools/libatom_hip.so ATOM_GEMVQ_EARLY_STREAM=1
run ATOM_LIB=$PWD/build/tools/libatom_hip.so ATOM_GEMVQ_EARLY_STREAM=2
run ATOM_LIB=$PWD/build/tools/libatom_hip.so ATOM_GEMVQ_EARLY_STREAM=3
done
} | tee $O/ab_gemvq_early_stream.txt
