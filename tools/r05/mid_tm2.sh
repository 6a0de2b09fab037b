#!/bin/bash
# Round 5, step 5: the BF6 mid-size-batch kernel on 128 x 64 tiles (TM = 2) against 64 x 64 (ATOM_MID_TM=1) and against the picked kernel
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05; mkdir -p $O
B=build/tools/gemm_bench
{
  for s in "512 4096 4096" "300 4096 4224" "384 2112 1408" "1024 4096 4096" "129 8192 1152" "512 4096 384"; do
    ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_TM=2 $B $s 5 100000 | grep "check"
    ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_TM=2 ATOM_NO_PAIRS=1 $B $s 5 100000 | grep "check"
  done
  for s in "256 4096 4096" "384 4096 4096" "512 4096 4096" "768 4096 4096" "1024 4096 4096" "256 11008 4096" "512 11008 4096" "512 4096 11008" "256 13824 5120" "512 5120 5120" "128 13824 5120" "2048 4096 4096"; do
    echo "== $s"
    echo -n "f6 picked  : "; ATOM_F6=1 $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    for tm in 1 2; do
      echo -n "f6 mid tm$tm : "; ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_TM=$tm $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    done
  done
  timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_gateup.py tests/test_gpu_block.py -m gpu -x -q -k "mid or two_k_group or gate_up or random_shapes or teacher or contract or f6" 2>&1 | tail -5
} > $O/mid_tm2.txt 2>&1
cat $O/mid_tm2.txt
