# PMC of the fused activation quantisers (run on the GPU box through gpurun): usage  tools/pmc_quant.sh TAG M H
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; M=${2:-4096}; H=${3:-4096}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$n -o q -- $R/build/tools/quant_bench $M $H 3 > $OUT/$n.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        name = 'rmsnorm' if 'ELi1E' in k and 'act_quant2' in k else ('reorder' if 'act_quant2' in k else ('silu' if 'silu_quant2' in k else None))
        if name: acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
print("PMC $TAG M=$M H=$H")
for kn,d in sorted(acc.items()):
    print(kn)
    for k,v in sorted(d.items()):
        print("  %-28s n=%d mean=%.4g"%(k,len(v),sum(v)/len(v)))
PY
