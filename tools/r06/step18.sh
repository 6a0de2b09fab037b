#!/bin/bash
# Round 6, step 18: decode attention -- nibbles widened through v_cvt_pk_f32_fp8 + plain FP32 FMAs instead of half2{1024 + n} + v_fma_mix_f32
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_e2e.py tests/test_gpu_block.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_step18.txt
timeout 600 python tools/cold_bench.py layer 1,8,16,64 2>&1 | grep "^batch" | tee $O/layer_step18.txt
timeout 600 bash tools/r06/decode_prof.sh step18_b16 16 2>&1 | grep batch_decode
timeout 600 bash tools/r06/decode_prof.sh step18_b1 1 2>&1 | grep "batch_decode\|merge"
