#!/bin/bash
# Round 5, step 3c: BF6 mid kernel with a ring of four unrolled (compile-time slots: no address arithmetic, no slot selection) and
# exact waits in the last steps (no repeated LDS-DMA, no drain ahead of the keeper)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05; mkdir -p $O
B=build/ab/new/gemm_bench
{
  for s in "64 4096 4096" "256 4096 4096" "300 4096 1152" "512 4096 4096" "100 13824 5120" "256 4096 384" "256 4096 256" "70 4096 640" "130 4096 896" "256 4096 1024"; do
    ATOM_F6=1 ATOM_F6_CFG=20 $B $s 5 100000 | grep "check"
    ATOM_F6=1 ATOM_F6_CFG=20 ATOM_NO_PAIRS=1 $B $s 5 100000 | grep "check"
    ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_NS=5 $B $s 5 100000 | grep "check"
  done
  for s in "64 4096 4096" "128 4096 4096" "256 4096 4096" "512 4096 4096" "1024 4096 4096" "64 13824 5120" "256 13824 5120" "64 5120 13824" "256 5120 13824" "128 11008 4096" "256 11008 4096" "256 4096 11008" "64 5120 5120" "256 5120 5120"; do
    echo "== $s"
    echo -n "f6 picked : "; ATOM_F6=1 $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    for ns in 4 5 6; do
      echo -n "f6 mid ns$ns: "; ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_NS=$ns $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    done
  done
  echo "## ablations, 256 x 4096 x 4096, ring of four (1 no LDS-DMA in the loop, 2 no barrier, 4 no MFMA, 8 no LDS loads, 16 no de-quantisation)"
  for a in 0 1 2 3 4 8 16 27 31; do
    echo -n "abl $a: "; ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_ABL=$a $B 256 4096 4096 300 0 | grep RESULT | sed 's/RESULT variant=default//'
  done
} > $O/mid_f6c.txt 2>&1
cat $O/mid_f6c.txt
