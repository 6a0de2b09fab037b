#!/bin/bash
# Round 6, step 20: the (non-persistent) dot-product kernel at 4 / 8 tokens (tools build carries those instances) against the decode-batch kernel
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 4,8 2>&1 | grep "^batch"; }
{
run ATOM_X=0
run ATOM_GEMV_TOKENS=4
run ATOM_GEMV_TOKENS=8
} | tee $O/ab_gemv_tokens.txt
