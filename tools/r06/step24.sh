#!/bin/bash
# Round 6, step 24: tile buffers per wave (2 .. 5) in the rebuilt decode attention; the layer by batch, same box
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
for d in 3 2 4 5 3; do
  echo "== tile buffers per wave: $d"
  ATOM_LIB=$PWD/build/ab/dp$d/libatom_hip.so timeout 300 python tools/cold_bench.py layer 1,16,64 2>&1 | grep "^batch"
done | tee $O/ab_decode_ring2.txt
