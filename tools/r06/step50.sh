#!/bin/bash
# Round 6, step 50: the half-share ring rule (default) against whole-share rings (ATOM_GEMVQ_DMAX=6), and o_proj's owning streamers with one slot
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "multi_q or fused_q or gemvq or quantiser" 2>&1 | tail -2
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1 2>&1 | grep "^batch"; }
{
run ATOM_GEMVQ_DMAX=6
run X=default
run ATOM_GEMVQ_OWN_D=1
run ATOM_GEMVQ_DMAX=6
run X=default
run ATOM_GEMVQ_OWN_D=1
} | tee $O/ab_gemvq_ring_rule.txt
