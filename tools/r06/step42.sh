#!/bin/bash
# Round 6, step 42: the merge-inside-o_proj launch with quantiser / streamer roles (the streamers own every feature, ring of 2)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -2
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1,2 2>&1 | grep "^batch"; }
{
run ATOM_MERGE_IN_O_PROJ=0
run ATOM_MERGE_IN_O_PROJ=1
run ATOM_MERGE_IN_O_PROJ=1 ATOM_GEMVQ_OWN=0
run ATOM_MERGE_IN_O_PROJ=0
run ATOM_MERGE_IN_O_PROJ=1
} | tee $O/ab_merge_in_o_proj3.txt
timeout 600 bash tools/r06/decode_prof.sh merge1 1 ATOM_MERGE_IN_O_PROJ=1 > /dev/null 2>&1; tail -9 $O/decode_prof_merge1.txt
