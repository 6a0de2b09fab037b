#!/bin/bash
# Round 6, step 12: gemvq with the ring sized to a wave's steps, requested unconditionally in step order, first trip straight-line (precise vmcnt).
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_step12.txt
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
{
echo "== early ring for SiLU x up (default)"
timeout 600 python tools/cold_bench.py layer 1,2 2>&1 | grep "^batch"
echo "== ATOM_GEMVQ_EARLY4=0"
ATOM_GEMVQ_EARLY4=0 timeout 600 python tools/cold_bench.py layer 1,2 2>&1 | grep "^batch"
echo "== early ring again"
timeout 600 python tools/cold_bench.py layer 1,2 2>&1 | grep "^batch"
} | tee $O/ab_gemvq_first_trip.txt
timeout 300 python tools/r06/gemvq_trace.py 2>&1 | tee $O/gemvq_trace_step12.txt | cut -c1-260 | grep -A3 "==" | head -60
