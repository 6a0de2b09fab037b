#!/bin/bash
# Round 5, step 3: the BF6-operand form of the mid-size-batch kernel (ATOM_F6_CFG=20 forces it in the tools build) against the kernels
# f6_pick_cfg runs for the same operands, and the packed-format kernel with one scale piece per stage instead of eight.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05; mkdir -p $O
B=build/ab/new/gemm_bench
{
  echo "## correctness (gemm_bench check vs FP64, all rows): BF6 mid kernel, pairs / no pairs, ring depths"
  for s in "64 4096 4096" "256 4096 4096" "300 4096 1152" "512 4096 4096" "100 13824 5120" "256 4096 384"; do
    ATOM_F6=1 ATOM_F6_CFG=20 $B $s 5 100000 | grep "check"
    ATOM_F6=1 ATOM_F6_CFG=20 ATOM_NO_PAIRS=1 ATOM_MID_NS=5 $B $s 5 100000 | grep "check"
  done
  for s in "64 4096 4096" "128 4096 4096" "256 4096 4096" "512 4096 4096" "1024 4096 4096" "64 13824 5120" "256 13824 5120" \
           "64 5120 13824" "256 5120 13824" "128 11008 4096" "256 11008 4096" "512 11008 4096" "256 4096 11008"; do
    echo "== $s"
    echo -n "f6 picked : "; ATOM_F6=1 $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    for ns in 10 8 5; do
      echo -n "f6 mid ns$ns: "; ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_NS=$ns $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    done
    echo -n "f6 mid nopair: "; ATOM_NO_PAIRS=1 ATOM_F6=1 ATOM_F6_CFG=20 $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    echo -n "packed mid 8w8s: "; ATOM_MID_MIN_TILES=1 ATOM_MID_MAX_M=100000 ATOM_MID_NW=8 ATOM_MID_NS=8 $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
  done
} > $O/mid_f6.txt 2>&1
cat $O/mid_f6.txt
