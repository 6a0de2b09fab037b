# two-K-group q-step kernel: parity tests, then same-box A/B against the x16 body (ATOM_QK=0) on the tools build, then the trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "two_k_group" 2>&1 | tail -5
for rep in 1 2; do
  for qk in 1 0; do
    for shape in "1024 4096 4096" "768 4096 4096" "1024 4096 11008" "1024 5120 5120"; do
      ATOM_QK=$qk ATOM_F6=1 timeout 120 build/tools/gemm_bench $shape 300 64 2>&1 | grep -E "RESULT|FAIL" | sed "s/^/qk=$qk rep$rep: /" | cut -c1-150
    done
  done
done 2>&1 | tee gpurun_out/r03/qk_ab.txt
for abl in 1 14 15; do
  ATOM_F6_CFG=$((2100+abl)) ATOM_F6=1 timeout 120 build/tools/gemm_bench 1024 4096 4096 300 0 2>&1 | grep -E "RESULT" | sed "s/^/abl=$abl: /" | cut -c1-120
done 2>&1 | tee gpurun_out/r03/qk_abl3.txt
timeout 120 build/tools/trace_f6q 1024 4096 4096 qk 2>&1 | tee gpurun_out/r03/trace_f6qk.txt | cut -c1-330 | grep -E "traced|K steps|group 0 wave 0:|group 1 wave 0:"
