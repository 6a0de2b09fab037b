#!/bin/bash
# Round 6, step 49: ring depth of the one-token launches re-measured on the final kernels (ATOM_GEMVQ_DMAX caps the slots; a wave's steps:
# q / k / v 3, gate / up 6, o_proj 2 with owning streamers)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1 2>&1 | grep "^batch"; }
{
run ATOM_GEMVQ_DMAX=6
run ATOM_GEMVQ_DMAX=3
run ATOM_GEMVQ_DMAX=2
run ATOM_GEMVQ_DMAX=1
run ATOM_GEMVQ_DMAX=6
run ATOM_GEMVQ_DMAX=3
run ATOM_GEMVQ_DMAX=2
run ATOM_GEMVQ_DMAX=1
} | tee $O/ab_gemvq_ring_depth2.txt
for d in 3 2; do timeout 600 bash tools/r06/decode_prof.sh dmax$d 1 ATOM_GEMVQ_DMAX=$d > /dev/null 2>&1; tail -9 $O/decode_prof_dmax$d.txt | cut -c1-140; done
