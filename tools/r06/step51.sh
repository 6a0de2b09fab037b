#!/bin/bash
# Round 6, step 51: SiLU x up -> down_proj: the weight ring (one feature = four parts per wave) requested in front of the codes (1, shipped),
# half in front and half behind (2), all behind (0)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -1
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1 2>&1 | grep "^batch"; }
{
run ATOM_GEMVQ_EARLY4=1
run ATOM_GEMVQ_EARLY4=2
run ATOM_GEMVQ_EARLY4=0
run ATOM_GEMVQ_EARLY4=1
run ATOM_GEMVQ_EARLY4=2
run ATOM_GEMVQ_EARLY4=0
} | tee $O/ab_gemvq_silu_half_ring.txt
