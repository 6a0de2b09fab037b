#!/bin/bash
# Round 6, step 16: two tokens -- the dot-product kernel (quantiser in front) for every projection against the K > 4096 rule.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 2 2>&1 | grep "^batch"; }
{
run ATOM_X=0
run ATOM_GEMV_TOKENS=2 ATOM_FUSED_Q_MASK2=15
run ATOM_GEMV_TOKENS=2 ATOM_FUSED_Q_MASK2=2
run ATOM_FUSED_Q_MASK2=10
run ATOM_FUSED_Q_MASK2=15
run ATOM_X=0
} | tee $O/ab_two_tokens.txt
