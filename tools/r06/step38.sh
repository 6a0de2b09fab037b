#!/bin/bash
# Round 6, step 38: SiLU x up -> down_proj with quantiser / streamer roles (the streamers own every feature)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -2
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1 2>&1 | grep "^batch"; }
{
run ATOM_GEMVQ_ROLES4=1
run ATOM_GEMVQ_ROLES4=0
run ATOM_GEMVQ_ROLES4=1
run ATOM_GEMVQ_ROLES4=0
} | tee $O/ab_gemvq_roles4.txt
timeout 300 python tools/r06/gemvq_trace.py 2>&1 | grep -v amdgpu | cut -c1-250 | grep -A6 "silu_mul"
