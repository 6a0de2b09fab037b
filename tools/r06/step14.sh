#!/bin/bash
# Round 6, step 14: the decode-batch kernel looping over its feature blocks at 3 .. 16 tokens (activations read once per workgroup):
# tests, then the layer by grid cap (0 = one workgroup per 16 features, the form of rounds 3-5).
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_step14.txt
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 4,8,16 2>&1 | grep "^batch"; }
{
run ATOM_SKINNY_PER_GRID=0
run ATOM_SKINNY_PER_GRID=256
run ATOM_SKINNY_PER_GRID=384
run ATOM_SKINNY_PER_GRID=512
run ATOM_SKINNY_PER_GRID=768
run ATOM_SKINNY_PER_GRID=0
} | tee $O/ab_skinny_persistent.txt
