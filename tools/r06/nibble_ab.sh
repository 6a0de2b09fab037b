#!/bin/bash
# Round 6: the price of keeping the weight at 4 bits in HBM on the prefill path -- QLinearLayer.keep_f6 = False re-codes
# the packed weight to BF6 into a transient buffer by every call -- against the cached BF6 form, same box: a Llama-7B block at 2, 8 and 32 x 2048
# tokens (tools/block_bench.py), and the bare GEMM through the C ABI (gemm_bench: BF6 operands | packed + workspace, nothing cached).
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_gpu_block.py -m gpu -q -x -k "memory_policy" 2>&1 | tail -2
{
  for b in 2 8 32; do
    for mode in f6 nibble; do
      echo "== block batch $b x 2048 tokens, weights: $mode"
      python tools/block_bench.py $b $mode 2>&1 | grep -E "BLOCK|seven W4A4|reference's inputs"
    done
  done
  for shape in "4096 4096 4096" "2048 11008 4096" "16384 4096 4096"; do
    echo "== GEMM $shape: BF6 operands"; ATOM_F6=1 build/tools/gemm_bench $shape 200 0 | grep RESULT
    echo "== GEMM $shape: packed operands + workspace, both re-coded per call"; ATOM_WS=1 build/tools/gemm_bench $shape 200 0 | grep RESULT
    echo "== GEMM $shape: packed operands + workspace, weight's BF6 form cached"; ATOM_WS=1 ATOM_WS_CACHED=1 build/tools/gemm_bench $shape 200 0 | grep RESULT
  done
} > $O/ab_nibble_weights.txt 2>&1
cat $O/ab_nibble_weights.txt
