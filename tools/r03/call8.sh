cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/prof_layer -o layer -- python $GRAFT_REPO_ROOT/tools/cold_bench.py layer 1 > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_layer.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/r03/prof_layer.log
f=$(find gpurun_out/r03/prof_layer -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:30]:
    print("%-110s calls %6s avg %8.2f us  %5s%%"%(r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
