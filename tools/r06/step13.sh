#!/bin/bash
# Round 6, step 13: gemvq counters polled without / with s_sleep (ATOM_GEMVQ_ROLES=5).
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1 2>&1 | grep "^batch"; }
{
run ATOM_GEMVQ_ROLES=1
run ATOM_GEMVQ_ROLES=5
run ATOM_GEMVQ_ROLES=1
run ATOM_GEMVQ_ROLES=5
} | tee $O/ab_gemvq_poll.txt
