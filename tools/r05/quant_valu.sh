#!/bin/bash
# Round 5: the RMSNorm quantisers with the exact reciprocal root without range handling (rinv_sqrt_exact) and the hidden-4096 instances
# (row buffers at compile-time LDS offsets): parity, the exhaustive probe of the reciprocal root, same-box A/B against HEAD before
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05; mkdir -p $O
{
  timeout 120 build/tools/rinv_probe
  timeout 1200 python -m pytest tests/test_gpu_quant.py tests/test_gpu_ref.py tests/test_gpu_e2e.py tests/test_gpu_gemm.py -m gpu -x -q -k "quant or ref or e2e or quantiser or multi_q" 2>&1 | tail -4
  for shape in "1024 4096" "4096 4096" "65536 4096" "4096 8192"; do
    for v in qold new; do
      if [ $v = new ]; then B=build/tools/quant_bench; else B=build/ab/qold/quant_bench; fi
      for rep in 1 2; do
        echo "== $v M H = $shape"
        timeout 120 $B $shape 200 2>&1 | grep -E "dequant_out=0" | grep -E "reorder|rmsnorm"
      done
    done
  done
  echo "== BF6 records 4096 x 4096"
  for v in qold new; do
    if [ $v = new ]; then B=build/tools/quant_bench; else B=build/ab/qold/quant_bench; fi
    echo "-- $v"; ATOM_QB_FMT=512 timeout 120 $B 4096 4096 200 2>&1 | grep -E "dequant_out=0" | grep -E "reorder|rmsnorm"
  done
} > $O/quant_valu.txt 2>&1
cat $O/quant_valu.txt
