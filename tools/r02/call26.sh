cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/pmc_q
{ for fmt in 0 512; do for M in 4096 65536; do echo "## fmt=$fmt M=$M"; ATOM_QB_FMT=$fmt timeout 120 build/tools/quant_bench $M 4096 $([ $M = 4096 ] && echo 200 || echo 20); done; done; } > gpurun_out/r02/quant_bench.txt 2>&1
cat gpurun_out/r02/quant_bench.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  ATOM_QB_FMT=512 timeout 150 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_$n -o p -- $GRAFT_REPO_ROOT/build/tools/quant_bench 65536 4096 3 > /tmp/pmc_$n.log 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r02/pmc_q/$n.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
  cat $GRAFT_REPO_ROOT/gpurun_out/r02/pmc_q/$n.txt 2>/dev/null | cut -c1-300
done
