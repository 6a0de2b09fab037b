#!/bin/bash
# Round 6, step 4: whole GPU suite, then the decode layer hot / cold (profiles/r06/decode_layer_hot_cold.txt) and its per-kernel view at
# batch 1 and 16 (profiles/r06/decode_prof_*.txt), and the quantiser-in-front kernel's s_memtime trace.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_step4.txt 2>&1; echo "pytest rc $?" >> $O/pytest_step4.txt; tail -5 $O/pytest_step4.txt
python tools/cold_bench.py layer 1,2,4,16,64 > $O/decode_layer_hot_cold.txt 2>&1; cat $O/decode_layer_hot_cold.txt
bash tools/r06/decode_prof.sh final_b1 1
bash tools/r06/decode_prof.sh final_b16 16
ATOM_LIB=build/tools/libatom_hip.so python tools/r06/gemvq_trace.py > $O/gemvq_trace.txt 2>&1
