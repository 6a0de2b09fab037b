export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$R/build/tools; O=$R/gpurun_out/c8; mkdir -p $O; cd $R
export ATOM_F6=1
for rep in 1 2; do for cfg in 0 30; do for shape in "4096 4096 4096"; do
  echo -n "cfg $cfg (0 = q, 30 = p) "; ATOM_F6_CFG=$cfg timeout 120 $T/gemm_bench $shape 300 256 | grep -E "check|RESULT" | tr '\n' ' '; echo; done; done; done > $O/ab.txt 2>&1
for shape in "1000 11008 4096" "8192 8192 8192" "300 512 384" "256 256 256" "4096 4096 512" "777 1024 640"; do for cfg in 0 30; do
  echo -n "cfg $cfg $shape: "; ATOM_F6_CFG=$cfg timeout 120 $T/gemm_bench $shape 100 100000 | grep -E "check|RESULT" | tr '\n' ' '; echo; done; done > $O/shapes.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
cat $O/ab.txt $O/shapes.txt $O/pytest.txt; cut -c1-600 $O/bench_driver.json; tail -3 $O/bench_driver.err
