# PMC pass over the INT4 decode-attention kernel (run on the GPU box through gpurun).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_decode
rm -rf $OUT; mkdir -p $OUT
cd /tmp
cat > /tmp/dec_one.py <<PY
import sys; sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import decode_bench as D
D.run(32, 2048, iters=20)
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$n -o d -- python /tmp/dec_one.py > $OUT/$n.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'batch_decode_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print("PMC decode batch=32 seqlen=2048")
for k,v in sorted(acc.items()):
    print("  %-24s n=%d mean=%.4g"%(k,len(v),sum(v)/len(v)))
PY
