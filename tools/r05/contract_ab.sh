#!/bin/bash
# Round 5, step 1 (one gpurun call): the whole GPU suite on the new arithmetic contract, then same-box A/B of the headline:
#   base = the kernels of round 4 (git HEAD at the start of the round), new = the working tree; `nopairs` = the new library without
#   ATOM_B_SCALE_PAIRS (8 instead of 6 de-quantisation VALU per MFMA).  build/ab/* come from tools/ab_build.sh.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_contract.txt 2>&1; echo "pytest rc $?" >> $O/pytest_contract.txt
tail -5 $O/pytest_contract.txt
build/tools/smfma_probe > $O/smfma_probe.txt 2>&1; cat $O/smfma_probe.txt
{
  for r in 1 2 3; do
    for v in base new; do
      echo "== $v f6 4096^3 (round $r)"; ATOM_F6=1 build/ab/$v/gemm_bench 4096 4096 4096 400 0 | grep RESULT
    done
    echo "== new nopairs f6 4096^3 (round $r)"; ATOM_NO_PAIRS=1 ATOM_F6=1 build/ab/new/gemm_bench 4096 4096 4096 400 0 | grep RESULT
  done
  for v in base new; do
    echo "== $v f6 8192^3"; ATOM_F6=1 build/ab/$v/gemm_bench 8192 8192 8192 60 0 | grep RESULT
    echo "== $v packed (INT8 tile kernel) 4096^3"; build/ab/$v/gemm_bench 4096 4096 4096 200 0 | grep RESULT
    echo "== $v packed ws route 4096^3"; ATOM_WS=1 build/ab/$v/gemm_bench 4096 4096 4096 200 0 | grep RESULT
    for m in 64 256; do echo "== $v packed M=$m 4096x4096"; build/ab/$v/gemm_bench $m 4096 4096 400 0 | grep RESULT; done
    for m in 512 1024 2048; do echo "== $v f6 M=$m 4096x4096"; ATOM_F6=1 build/ab/$v/gemm_bench $m 4096 4096 400 0 | grep RESULT; done
    echo "== $v f6 2048x11008x4096"; ATOM_F6=1 build/ab/$v/gemm_bench 2048 11008 4096 200 0 | grep RESULT
  done
} > $O/ab_contract.txt 2>&1
cat $O/ab_contract.txt
