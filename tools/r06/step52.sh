#!/bin/bash
# Round 6, step 52: q / k / v with two thirds of a workgroup's features on the streamer waves (ring of two), re-measured on the half-share rings
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
ATOM_GEMVQ_SPLIT=1 timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -1
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1 2>&1 | grep "^batch"; }
{
run ATOM_GEMVQ_SPLIT=0
run ATOM_GEMVQ_SPLIT=1
run ATOM_GEMVQ_SPLIT=0
run ATOM_GEMVQ_SPLIT=1
} | tee $O/ab_gemvq_split2.txt
ATOM_GEMVQ_SPLIT=1 timeout 300 python tools/r06/gemvq_trace.py 2>&1 | grep -v amdgpu | cut -c1-250 | grep -A6 "== rmsnorm"
