# quantisers inside the decode GEMMs (atom_gemm_w4a4_multi_q): parity tests, stand-alone timing, the decode layer by fusion mask
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_gemm.py -x -q -m gpu -k "quantiser_inside or quantisers_inside or decode" 2>&1 | tail -3
timeout 300 python tools/r03/fq_probe.py 2>&1 | grep -E "M=" | tee gpurun_out/r03/fq_probe.txt
for rep in 1 2; do
  for mask in 0 2 15; do
    ATOM_FUSED_Q_MASK=$mask timeout 300 python tools/cold_bench.py layer 1,2 2>&1 | grep batch | sed "s/^/mask=$mask rep$rep: /"
  done
done 2>&1 | tee gpurun_out/r03/fused_q_layer.txt
