#!/bin/bash
# Round 6, step 3: the whole GPU suite on the gemvq dispatch (one / two tokens: quantiser in front of the dot-product kernel), then the decode
# layer hot / cold at batch 1, 2, 4, 16 and its per-kernel view.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_step3.txt 2>&1; echo "pytest rc $?" >> $O/pytest_step3.txt; tail -5 $O/pytest_step3.txt
python tools/cold_bench.py layer 1,2,4,16 > $O/decode_layer_step3.txt 2>&1; cat $O/decode_layer_step3.txt
bash tools/r06/decode_prof.sh s3b1 1
bash tools/r06/decode_prof.sh s3b2 2
