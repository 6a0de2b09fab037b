#!/bin/bash
# Round 6, step 40: the KV-split merge inside o_proj's launch, re-measured now that the attention leaves 4 partial states per head at batch 1, context 1024
# (two-level merge) instead of 16-32.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer 1,2 2>&1 | grep "^batch"; }
{
run ATOM_MERGE_IN_O_PROJ=0
run ATOM_MERGE_IN_O_PROJ=1
run ATOM_MERGE_IN_O_PROJ=0
run ATOM_MERGE_IN_O_PROJ=1
} | tee $O/ab_merge_in_o_proj2.txt
