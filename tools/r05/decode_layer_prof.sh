#!/bin/bash
# Round 5: per-kernel view of one decode step of a Llama-7B layer at batch 1 (rocprofv3 kernel trace of tools/cold_bench.py layer 1)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp
python $R/tools/cold_bench.py layer 1 > $O/decode_layer1.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dl -o dl -- python $R/tools/cold_bench.py layer 1 > /tmp/dl.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/dl/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
with open("$O/decode_layer1_kernel_stats.csv","w") as o:
    w=csv.writer(o); w.writerow(["Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs"])
    for r in rows[:24]: w.writerow([r["Name"][:150],r["Calls"],r["TotalDurationNs"],r["AverageNs"],r["Percentage"],r["MinNs"]])
PY
cat $O/decode_layer1.txt; cut -c1-190 $O/decode_layer1_kernel_stats.csv
