# PMC comparison of two gemm_bench configurations (run on the GPU box through gpurun): usage  tools/pmc_gemm.sh TAG [env...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  n=$(echo $set | cut -d' ' -f1)
  env "$@" rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$n -o g -- $R/build/tools/gemm_bench 4096 4096 4096 40 0 > $OUT/$n.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'gemm_w4a4' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print("PMC $TAG")
for k,v in sorted(acc.items()):
    print("  %-28s n=%d mean=%.4g"%(k,len(v),sum(v)/len(v)))
PY
