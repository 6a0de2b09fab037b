export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$R/build/tools; O=$R/gpurun_out/c3; mkdir -p $O; cd $R
export ATOM_F6=1
for cfg in 0 40 3; do for shape in "4096 4096 4096" "1000 11008 4096" "2048 4096 4096"; do
  echo "cfg $cfg shape $shape"; ATOM_F6_CFG=$cfg timeout 120 $T/gemm_bench $shape 300 256 | grep -E "check|RESULT"; done; done > $O/bench_shapes.txt 2>&1
for c in 0 1001 1002 1008 1003 1010 1015 1207; do echo -n "cfg $c "; ATOM_F6_CFG=$c timeout 60 $T/gemm_bench 4096 4096 4096 300 0 | grep RESULT; done > $O/abl.txt 2>&1
timeout 60 $T/trace_f6p > $O/trace.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest.txt
cat $O/bench_shapes.txt $O/abl.txt $O/pytest.txt; grep -E "wave [04]:|traced step  [1357]:" $O/trace.txt | head -12
