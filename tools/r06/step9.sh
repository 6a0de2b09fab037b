#!/bin/bash
# Round 6, step 9: gemvq with the streamer waves joining the quantiser for the codes: e2e / gemm tests, layer, trace.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_gemm.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_step9.txt
{
echo "== roles 1 (default)"
timeout 600 python tools/cold_bench.py layer 1,2 2>&1 | grep "^batch"
echo "== roles 0 (tools build)"
ATOM_LIB=$PWD/build/tools/libatom_hip.so ATOM_GEMVQ_ROLES=0 timeout 600 python tools/cold_bench.py layer 1 2>&1 | grep "^batch"
} | tee $O/ab_gemvq_codes16.txt
timeout 300 python tools/r06/gemvq_trace.py 2>&1 | tee $O/gemvq_trace_codes16.txt | cut -c1-260 | head -60
