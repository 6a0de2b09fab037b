#!/bin/bash
# Round 6, step 5: the rocprofv3 passes of the headline command (tools/profile_bench.sh -> gpurun_out/prof_summary), the s_memtime trace of
# the headline kernel incl. the split of one K step, and the default bench line.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
build/tools/trace_f6q 4096 4096 4096 > $O/trace_f6q.txt 2>&1; tail -12 $O/trace_f6q.txt | cut -c1-400
bash tools/profile_bench.sh > $O/profile_bench.log 2>&1; tail -5 $O/profile_bench.log
ls gpurun_out/prof_summary
python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 600 $O/bench_line.json
