export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$R/build/tools; O=$R/gpurun_out/c18; mkdir -p $O; cd $R
export ATOM_F6=1
for rep in 1 2 3; do for cfg in 0 30; do echo -n "cfg $cfg "; ATOM_F6_CFG=$cfg timeout 120 $T/gemm_bench 4096 4096 4096 300 $([ $rep = 1 ] && echo 256 || echo 0) | grep -E "check|RESULT" | tr '\n' ' '; echo; done; done > $O/ab.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_gateup.py -m gpu -q -x 2>&1 | tail -3 >> $O/ab.txt
cat $O/ab.txt
