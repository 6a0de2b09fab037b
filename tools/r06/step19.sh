#!/bin/bash
# Round 6, step 19 (TIMING ONLY, tools build): does the decode-batch kernel's weight access pattern -- 16 rows x 64 bytes per load
# instruction, consecutive lanes on different rows -- limit its stream?  NB = 3 blocks per workgroup (all loads up front) with the lanes
# re-mapped to consecutive bytes (results are garbage: only the kernel time is read).
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$PWD/build/tools/libatom_hip.so
for c in 0 1 2; do
  echo "== ATOM_SKINNY_NB=3 ATOM_SKINNY_COAL=$c"
  timeout 600 bash tools/r06/decode_prof.sh coal${c}_b16 16 ATOM_SKINNY_NB=3 ATOM_SKINNY_COAL=$c 2>&1 | grep "skinny_nb\|^batch"
done | tee $O/ab_skinny_coalesced_timing.txt
