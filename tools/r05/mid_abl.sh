#!/bin/bash
# Round 5: where a K step of the BF6 mid kernel goes (256 x 4096 x 4096, ring of 5; ablation bits: 1 no LDS-DMA in the loop, 2 no
# barrier, 4 no MFMA, 8 no LDS loads, 16 no de-quantisation -- results are wrong by construction, timing only)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05; mkdir -p $O
B=build/ab/new/gemm_bench
{
  for a in 0 1 2 3 4 8 16 20 28 9 11 27 31; do
    echo -n "abl $a: "; ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_NS=5 ATOM_MID_ABL=$a $B 256 4096 4096 300 0 | grep RESULT | sed 's/RESULT variant=default//'
  done
} > $O/mid_abl.txt 2>&1
cat $O/mid_abl.txt
