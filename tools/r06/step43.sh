#!/bin/bash
# Round 6, step 43: s_memtime trace of the merge-inside-o_proj launch (roles, streamers own every feature) next to the plain reorder launch
cd "$(dirname "$0")/../.." || exit 1
TRACE_MERGE=1 ATOM_LIB=$PWD/build/tools/libatom_hip.so timeout 300 python tools/r06/gemvq_trace.py 2>&1 | grep -v amdgpu | cut -c1-260 | grep -A7 "== reorder\|== merge"
