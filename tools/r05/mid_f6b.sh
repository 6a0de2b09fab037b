#!/bin/bash
# Round 5, step 3b: BF6 mid kernel after (1) scales / accumulators in two alternating register sets (no exposed LDS latency, no copies)
# and (2) the LDS-DMA issued behind the step's LDS loads.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05; mkdir -p $O
B=build/ab/new/gemm_bench
{
  for s in "64 4096 4096" "256 4096 4096" "300 4096 1152" "512 4096 4096" "100 13824 5120" "256 4096 384" "256 4096 256" "70 4096 640"; do
    ATOM_F6=1 ATOM_F6_CFG=20 $B $s 5 100000 | grep "check"
    ATOM_F6=1 ATOM_F6_CFG=20 ATOM_NO_PAIRS=1 ATOM_MID_NS=8 $B $s 5 100000 | grep "check"
  done
  for s in "64 4096 4096" "256 4096 4096" "512 4096 4096" "1024 4096 4096" "64 13824 5120" "256 13824 5120" "256 5120 13824" "256 11008 4096" "256 4096 11008"; do
    echo "== $s"
    echo -n "f6 picked : "; ATOM_F6=1 $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    for ns in 10 8 5; do
      echo -n "f6 mid ns$ns: "; ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_NS=$ns $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    done
  done
} > $O/mid_f6b.txt 2>&1
cat $O/mid_f6b.txt
