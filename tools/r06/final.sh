#!/bin/bash
# Round 6 close-out on the GPU box: the whole GPU suite, the decode layer by batch, its per-kernel split at batch 1 and 16, the s_memtime
# traces, the rocprofv3 passes of the headline command (-> gpurun_out/prof_summary) and the default bench line.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee $O/pytest_gpu_final.txt
timeout 900 python tools/cold_bench.py layer 1,2,4,8,16,32,64 2>&1 | grep -v amdgpu.ids | tee $O/decode_layer_hot_cold.txt
timeout 600 bash tools/r06/decode_prof.sh final_b1 1 > /dev/null 2>&1; tail -9 $O/decode_prof_final_b1.txt
timeout 600 bash tools/r06/decode_prof.sh final_b16 16 > /dev/null 2>&1; tail -12 $O/decode_prof_final_b16.txt
ATOM_LIB=$PWD/build/tools/libatom_hip.so timeout 300 python tools/r06/gemvq_trace.py 2>&1 | grep -v amdgpu.ids > $O/gemvq_trace_final.txt
build/tools/trace_f6q 4096 4096 4096 > $O/trace_f6q.txt 2>&1; head -3 $O/trace_f6q.txt | cut -c1-200
bash tools/profile_bench.sh > $O/profile_bench.log 2>&1; tail -3 $O/profile_bench.log
ls gpurun_out/prof_summary
python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 1500 $O/bench_line.json
