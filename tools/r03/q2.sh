cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "256x128" 2>&1 | tail -5
for rep in 1 2; do
  for q2 in 1 0; do
    for shape in "2048 4096 4096" "1536 4096 4096" "1280 4096 4096" "2048 4096 11008" "512 13824 5120" "1024 8192 4096"; do
      ATOM_Q2=$q2 ATOM_F6=1 timeout 120 build/tools/gemm_bench $shape 300 64 2>&1 | grep -E "RESULT|FAIL" | sed "s/^/q2=$q2 rep$rep: /" | cut -c1-150
    done
  done
done 2>&1 | tee gpurun_out/r03/q2_ab.txt
