export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$R/build/tools; O=$R/gpurun_out/c9; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/pytest.txt
( echo -n "packed plain: "; timeout 120 $T/gemm_bench 4096 4096 4096 200 0 | grep RESULT
  echo -n "packed ws:    "; ATOM_WS=1 timeout 120 $T/gemm_bench 4096 4096 4096 200 0 | grep RESULT
  echo -n "wide:         "; ATOM_AWIDE=1 timeout 120 $T/gemm_bench 4096 4096 4096 200 0 | grep RESULT
  echo -n "f6 cfg3 2048: "; ATOM_F6=1 ATOM_F6_CFG=3 timeout 120 $T/gemm_bench 2048 4096 4096 200 0 | grep RESULT
  echo -n "f6 q:         "; ATOM_F6=1 timeout 120 $T/gemm_bench 4096 4096 4096 200 0 | grep RESULT
  echo -n "packed 1024:  "; ATOM_WS=1 timeout 120 $T/gemm_bench 1024 4096 4096 200 0 | grep RESULT
) > $O/gemm.txt 2>&1
for op in 0 1 2; do timeout 60 $T/quant_bench $op 4096 4096 2>&1 | tail -2; done > $O/quant.txt 2>&1
cat $O/pytest.txt $O/gemm.txt $O/quant.txt
