cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/prof2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for M in 512 1024; do
  for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    n=$(echo $set | cut -d" " -f1)
    ATOM_F6=1 timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_${M}_$n -o p -- $R/build/tools/gemm_bench $M 4096 4096 60 0 > /tmp/pmc_${M}_$n.log 2>&1
    f=$(find /tmp/pmc_${M}_$n -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" $M <<'PY' >> $R/gpurun_out/r02/prof2/midm_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "f6x16" not in k: continue
    k = k[:90]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print("M=%s" % sys.argv[2], k, {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()}, "launches", max(cnt[(k, c)] for c in acc[k]))
PY
  done
done
cat $R/gpurun_out/r02/prof2/midm_pmc.txt | cut -c1-400
