#!/bin/bash
# Round 6, step 57: the merge op requests only the partial states that exist (4 at context 1024: 8 requests per thread instead of 16)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -1
timeout 600 python tools/r06/gemvq_stress.py 2>&1 | grep merge
{
for r in 1 2 3; do for L in build/ab/head build/tools; do echo "== $L"; ATOM_LIB=$PWD/$L/libatom_hip.so timeout 300 python tools/cold_bench.py layer 1 2>&1 | grep "^batch"; done; done
} | tee $O/ab_merge_guarded_loads.txt
TRACE_MERGE=4 ATOM_LIB=$PWD/build/tools/libatom_hip.so timeout 300 python tools/r06/gemvq_trace.py 2>&1 | grep -v amdgpu | cut -c1-260 | grep -A2 "== merge"
