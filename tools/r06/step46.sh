#!/bin/bash
# Round 6, step 46 (timing only): the merge op reading 16-byte-aligned partial-state records (stride 132 floats) against the shipped 130
cd "$(dirname "$0")/../.." || exit 1
for r in 1 2; do
TRACE_MERGE=1 ATOM_LIB=$PWD/build/tools/libatom_hip.so timeout 300 python tools/r06/gemvq_trace.py 2>&1 | grep -v amdgpu | cut -c1-260 | grep -A3 "== merge"
TRACE_MERGE=1 PART_STRIDE=132 ATOM_LIB=$PWD/build/ab/pstr132/libatom_hip.so timeout 300 python tools/r06/gemvq_trace.py 2>&1 | grep -v amdgpu | cut -c1-260 | grep -A3 "== merge"
done
