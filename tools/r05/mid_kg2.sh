#!/bin/bash
# Round 5, step 4: the BF6 mid-size-batch kernel with TWO wave groups per tile (KG = 2: summation order 2) against one (ATOM_MID_KG=1)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05; mkdir -p $O
B=build/tools/gemm_bench
{
  for s in "64 4096 4096" "256 4096 4096" "200 4096 4224" "250 2112 1408" "100 13824 5120" "129 4096 1408"; do
    ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_KG=2 $B $s 5 100000 | grep "check"
    ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_KG=2 ATOM_NO_PAIRS=1 $B $s 5 100000 | grep "check"
  done
  for s in "64 4096 4096" "128 4096 4096" "192 4096 4096" "256 4096 4096" "64 13824 5120" "64 5120 13824" "128 11008 4096" "256 4096 11008" "64 5120 5120" "256 5120 5120" "256 4096 1408" "129 8192 8192"; do
    echo "== $s"
    for kg in 1 2; do
      echo -n "f6 mid kg$kg: "; ATOM_F6=1 ATOM_F6_CFG=20 ATOM_MID_KG=$kg $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    done
  done
  timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_gateup.py tests/test_gpu_block.py -m gpu -x -q -k "mid or two_k_group or gate_up or random_shapes or teacher or contract" 2>&1 | tail -8
} > $O/mid_kg2.txt 2>&1
cat $O/mid_kg2.txt
