#!/bin/bash
# Round 5, step 6: the BF6 mid-size-batch kernel on 128 x 128 tiles (ATOM_MID_TM=3) against what the dispatch picks today
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05; mkdir -p $O
B=build/tools/gemm_bench
{
  for s in "1024 4096 4096" "600 4096 4224" "768 2176 1408" "129 8192 1152" "1000 4096 384"; do
    ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_TM=3 $B $s 5 100000 | grep "check"
  done
  for s in "512 4096 4096" "768 4096 4096" "1024 4096 4096" "1536 4096 4096" "2048 4096 4096" "256 11008 4096" "512 11008 4096" "1024 11008 4096" "1024 4096 11008" "256 13824 5120" "512 13824 5120" "512 5120 5120" "1024 5120 5120" "4096 4096 4096"; do
    echo "== $s"
    echo -n "f6 picked  : "; ATOM_F6=1 $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    for tm in 2 3; do
      echo -n "f6 mid tm$tm : "; ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_TM=$tm $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    done
  done
} > $O/mid_tm3.txt 2>&1
cat $O/mid_tm3.txt
