#!/bin/bash
# Round 6, step 39: decode attention prologue -- (a) RoPE frequencies and the 16-position step computed once per wave (lane j: pair j),
# shared through wave-local LDS; (b) the seven run-time integer divisions by reciprocals the host prepared.  Bit identity against the previous build (build/ab/head), then the layer at batches 1 / 16 / 64.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
{
echo "== decode_fuzz digest, previous build / this build (tools libraries)"
ATOM_LIB=$PWD/build/ab/head/libatom_hip.so timeout 600 python tools/r06/decode_fuzz.py 60 2>&1 | tail -1
ATOM_LIB=$PWD/build/tools/libatom_hip.so timeout 600 python tools/r06/decode_fuzz.py 60 2>&1 | tail -1
for r in 1 2; do
for L in build/ab/head/libatom_hip.so build/tools/libatom_hip.so; do
  echo "== $L"
  for b in 1 16 64; do ATOM_LIB=$PWD/$L timeout 300 python tools/cold_bench.py layer $b 2>&1 | grep "^batch"; done
done; done
} 2>&1 | tee $O/ab_decode_rope_shared.txt
timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -2
