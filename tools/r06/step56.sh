#!/bin/bash
# Round 6, step 56: streamers owning EVERY feature (the quantiser waves exit after publishing) for q / k / v (6 steps per streamer, ring of 3)
# and gate / up (11 steps, ring of 3), re-measured on half-share rings
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
ATOM_GEMVQ_OWN_MAX=11 timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -1
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1 2>&1 | grep "^batch"; }
{
run ATOM_GEMVQ_OWN_MAX=2
run ATOM_GEMVQ_OWN_MAX=6
run ATOM_GEMVQ_OWN_MAX=11
run ATOM_GEMVQ_OWN_MAX=2
run ATOM_GEMVQ_OWN_MAX=6
run ATOM_GEMVQ_OWN_MAX=11
} | tee $O/ab_gemvq_own3.txt
