#!/bin/bash
# Round 6, step 54 (timing variant -DATOM_SKINNY_ACT_FIRST): decode-batch GEMM with the token block's activations requested in FRONT of the
# weights, so that a wave's first MFMA can start when its first weight chunk lands (in-order returns: with the activations behind the
# weights nothing of a wave computes before all of its weights have arrived)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
{
for s in "16 4096 4096" "16 4096 11008" "16 22016 4096" "16 12288 4096" "64 4096 11008" "4 4096 11008"; do
  echo "== $s"
  for L in tools ab/actfirst; do echo -n "$L: "; build/$L/gemm_bench $s 300 0 | grep RESULT | sed 's/RESULT variant=default//' | cut -c1-90; done
done
for r in 1 2; do for L in build/tools build/ab/actfirst; do echo "== $L"; ATOM_LIB=$PWD/$L/libatom_hip.so timeout 300 python tools/cold_bench.py layer 4,16 2>&1 | grep "^batch"; done; done
} 2>&1 | tee $O/ab_skinny_act_first.txt
