#!/bin/bash
# Round 6, step 41: where the merge-inside-o_proj layer spends its time: per-kernel durations and the gap in front of each launch
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 600 bash tools/r06/decode_prof.sh merge1 1 ATOM_MERGE_IN_O_PROJ=1 > /dev/null 2>&1; tail -9 $O/decode_prof_merge1.txt
timeout 600 bash tools/r06/decode_prof.sh merge0 1 ATOM_MERGE_IN_O_PROJ=0 > /dev/null 2>&1; tail -9 $O/decode_prof_merge0.txt
