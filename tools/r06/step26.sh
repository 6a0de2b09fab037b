#!/bin/bash
# Round 6, step 26 (TIMING ONLY): the headline kernel with its feature fragments read as 3 x ds_read_b128 per row PAIR -- what a
# pair-interleaved BF6 record would allow -- against the shipped 2 x 3 x ds_read_b64 (build/ab/pair128: -DATOM_F6_PAIR128; results garbage)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
{
for r in 1 2 3; do
  echo "== shipped (round $r)"; ATOM_F6=1 build/ab/base/gemm_bench 4096 4096 4096 400 0 | grep RESULT
  echo "== pair b128, timing only (round $r)"; ATOM_F6=1 build/ab/pair128/gemm_bench 4096 4096 4096 400 0 | grep RESULT
done
} 2>&1 | tee $O/ab_pair_b128_timing.txt
