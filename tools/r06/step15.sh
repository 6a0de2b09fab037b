#!/bin/bash
# Round 6, step 15: NB feature blocks per workgroup in the decode-batch kernel at 3 .. 16 tokens (activations read once per NB x 16 features)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_step15.txt
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 4,8,16 2>&1 | grep "^batch"; }
{
run ATOM_SKINNY_NB=0
run ATOM_SKINNY_NB=2
run ATOM_SKINNY_NB=3
run ATOM_SKINNY_NB=0
} | tee $O/ab_skinny_nb.txt
timeout 600 bash tools/r06/decode_prof.sh step15_b16 16 2>&1 | tail -12
