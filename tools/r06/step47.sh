#!/bin/bash
# Round 6, step 47: two tokens -- every quantiser fused on the dot-product kernel (ATOM_GEMV_TOKENS=2, q_mask2 15) against the shipped rule
# (K <= 4096 on the decode-batch kernel, only reorder -> o_proj fused), re-measured on the final kernels
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 2 2>&1 | grep "^batch"; }
{
run X=0
run ATOM_GEMV_TOKENS=2 ATOM_FUSED_Q_MASK2=15
run ATOM_GEMV_TOKENS=2 ATOM_FUSED_Q_MASK2=2
run X=0
run ATOM_GEMV_TOKENS=2 ATOM_FUSED_Q_MASK2=15
} | tee $O/ab_two_tokens2.txt
