#!/bin/bash
# Round 6, step 33: waves per workgroup of the two-level merge (2 / 4 / 8)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
for k in 4 2 8 4; do
  echo "== $k waves per workgroup"
  ATOM_LIB=$PWD/build/ab/inner$k/libatom_hip.so timeout 300 python tools/cold_bench.py layer 1,2,4 2>&1 | grep "^batch"
done | tee $O/ab_decode_inner_waves.txt
