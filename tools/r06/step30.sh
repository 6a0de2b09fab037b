#!/bin/bash
# Round 6, step 30: OWN (streamers own every feature) where the share is at most two steps per streamer wave (o_proj)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -2
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1 2>&1 | grep "^batch"; }
{
run ATOM_GEMVQ_OWN=1
run ATOM_GEMVQ_OWN=0
run ATOM_GEMVQ_OWN=1
run ATOM_GEMVQ_OWN=0
} | tee $O/ab_gemvq_own2.txt
