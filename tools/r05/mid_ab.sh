#!/bin/bash
# Round 5, step 2 (one gpurun call): the mid-size-batch kernel (gemm_w4a4_mid.hip) against the kernels the round-4 dispatch runs for the
# same packed operands, same box: build/ab/new = the working tree with -DATOM_TOOLS (ATOM_MID=0 restores the old dispatch;
# ATOM_MID_NW / ATOM_MID_NS force a geometry; ATOM_WS=1 = through atom_gemm_w4a4_f16_ws, i.e. the re-coding route above 256 rows).
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05; mkdir -p $O
B=build/ab/new/gemm_bench
{
  echo "## correctness of the mid kernel (gemm_bench check vs FP64 on all rows)"
  for s in "64 4096 4096" "256 4096 4096" "300 4096 1152" "512 4096 4096" "100 13824 5120"; do
    ATOM_MID_MIN_TILES=1 $B $s 5 100000 | grep "check"
    ATOM_MID_MIN_TILES=1 ATOM_NO_PAIRS=1 ATOM_MID_NW=4 ATOM_MID_NS=8 $B $s 5 100000 | grep "check"
  done
  echo "## timing: old dispatch (ATOM_MID=0; M > 256: also the ws route) | mid kernel default | forced geometries"
  for s in "32 4096 4096" "64 4096 4096" "128 4096 4096" "256 4096 4096" "512 4096 4096" "1024 4096 4096" "64 13824 5120" "256 13824 5120" \
           "64 5120 13824" "256 5120 13824" "64 5120 5120" "128 11008 4096" "256 11008 4096" "512 11008 4096" "256 4096 11008"; do
    echo "== $s"
    echo -n "old      : "; ATOM_MID=0 $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    echo -n "old ws   : "; ATOM_MID=0 ATOM_WS=1 $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    echo -n "old ws wc: "; ATOM_MID=0 ATOM_WS=1 ATOM_WS_CACHED=1 $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    echo -n "mid      : "; ATOM_MID_MIN_TILES=1 ATOM_MID_MAX_M=100000 $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    for g in "8 16" "8 8" "4 16" "4 8"; do
      set -- $g
      echo -n "mid $1w $2s: "; ATOM_MID_MIN_TILES=1 ATOM_MID_MAX_M=100000 ATOM_MID_NW=$1 ATOM_MID_NS=$2 $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
    done
    echo -n "mid nopair: "; ATOM_NO_PAIRS=1 ATOM_MID_MIN_TILES=1 ATOM_MID_MAX_M=100000 $B $s 300 0 | grep RESULT | sed 's/RESULT variant=default//'
  done
  echo "## the INT8 tile kernel of the plain entry with shared scale products (4096^3)"
  $B 4096 4096 4096 200 0 | grep RESULT
  ATOM_NO_PAIRS=1 $B 4096 4096 4096 200 0 | grep RESULT
} > $O/mid_ab.txt 2>&1
cat $O/mid_ab.txt
python -m pytest tests/test_gpu_gemm.py -x -q -k "mid or bit_exact_vs_c_contract or decode_batches" > $O/pytest_mid.txt 2>&1; tail -15 $O/pytest_mid.txt
