# PMC of one GEMM shape through build/tools/gemm_bench (run on the GPU box through gpurun): usage  tools/pmc_gemm.sh TAG M N K [env assignments...]
#   e.g.  tools/pmc_gemm.sh mid256 256 4096 4096 ATOM_F6=1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; M=$2; N=$3; K=$4; shift 4
for kv in "$@"; do export "$kv"; done
OUT=/tmp/pmc_gemm_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$n -o g -- $R/build/tools/gemm_bench $M $N $K 20 0 > $OUT/$n.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o g -- $R/build/tools/gemm_bench $M $N $K 200 0 > $OUT/stats.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'gemm' in k or 'repack' in k: acc[k[:110]][r['Counter_Name']].append(float(r['Counter_Value']))
print("PMC $TAG  M=$M N=$N K=$K  $*")
for kn,d in sorted(acc.items()):
    print(kn)
    for k,v in sorted(d.items()):
        print("  %-28s n=%d mean=%.4g"%(k,len(v),sum(v)/len(v)))
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:3]:
        print("stats: %-90s calls %s avg %.2f us" % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
