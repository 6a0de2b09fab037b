#!/bin/bash
# Round 6, step 59: a Llama-13B layer (hidden 5120: rows of 3-4 chunks per lane, a feature = two ring steps, no roles) -- ring of 6 / 4 / 2 steps
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so ATOM_LAYER_DIMS=5120,40,13824
run() { echo "== $*"; env "$@" timeout 400 python tools/cold_bench.py layer 1 2>&1 | grep "^batch"; }
{
run ATOM_GEMVQ_DMAX2=6
run ATOM_GEMVQ_DMAX2=4
run ATOM_GEMVQ_DMAX2=2
run ATOM_GEMVQ_DMAX2=6
run ATOM_GEMVQ_DMAX2=4
run ATOM_GEMVQ_DMAX2=2
} 2>&1 | tee $O/ab_gemvq_ring_depth_13b.txt
