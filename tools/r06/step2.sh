#!/bin/bash
# Round 6, step 2: the quantiser-in-front GEMM as a loop over feature blocks (one quantiser per workgroup), and the KV-split count of the
# decode attention at batch 1.  tools build (reads ATOM_* tuning variables) through ATOM_LIB.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_gpu_e2e.py -m gpu -x -q > $O/pytest_step2.txt 2>&1; echo "pytest rc $?" >> $O/pytest_step2.txt; tail -3 $O/pytest_step2.txt
export ATOM_LIB=build/tools/libatom_hip.so
{
  for mask in 2 15; do
    for grid in 256 100000; do
      echo "== q_mask $mask, quantiser-in-front grid cap $grid"
      ATOM_FUSED_Q_MASK=$mask ATOM_SKINNY_Q_GRID=$grid python tools/cold_bench.py layer 1,2 2>&1 | grep batch
    done
  done
  for mt in 8 4 2 1; do
    echo "== q_mask 15, grid 256, decode min tiles per split $mt"
    ATOM_FUSED_Q_MASK=15 ATOM_DECODE_MIN_TILES=$mt python tools/cold_bench.py layer 1,16 2>&1 | grep batch
  done
} > $O/decode_step2.txt 2>&1
cat $O/decode_step2.txt
