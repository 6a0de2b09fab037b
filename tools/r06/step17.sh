#!/bin/bash
# Round 6, step 17: roles / ring waits at TWO tokens; the dot-product kernel for every projection (tools: ATOM_GEMV_TOKENS=2) + all four
# quantisers fused against the K > 4096 rule.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_gemm.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_step17.txt
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer ${B:-2} 2>&1 | grep "^batch"; }
{
run ATOM_X=0
run ATOM_GEMV_TOKENS=2 ATOM_FUSED_Q_MASK2=15
run ATOM_GEMV_TOKENS=2 ATOM_FUSED_Q_MASK2=15 ATOM_GEMVQ_ROLES=0
run ATOM_FUSED_Q_MASK2=10
run ATOM_X=0
B=1 run ATOM_X=0
} | tee $O/ab_two_tokens_roles.txt
