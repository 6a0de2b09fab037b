# PMC of any gemm_bench invocation (run on the GPU box through gpurun): usage  tools/pmc_any.sh TAG KERNEL_SUBSTRING <gemm_bench args...>   (env passes through)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; PAT=$2; shift 2
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_SALU"; do
  # (SQ / GRBM counters only: a pass with TA_* counters hung rocprofv3 for its whole time limit on this pool)
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o g -- $R/build/tools/gemm_bench "$@" > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$PAT" in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print("PMC $TAG  kernel ~ $PAT  args: $*")
for k,v in sorted(acc.items()):
    print("  %-32s n=%d mean=%.5g"%(k,len(v),sum(v)/len(v)))
PY
rm -rf $OUT/p*/
