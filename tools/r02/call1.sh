# round 2, GPU call 1: GPU test suite, pipelined F6 kernel vs the first-generation one, ablations, trace, PMC, bench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$R/build/tools
O=$R/gpurun_out/c1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest.txt
export ATOM_F6=1
for cfg in 0 40; do for shape in "4096 4096 4096" "1000 11008 4096" "8192 8192 8192" "2048 4096 11008" "300 512 384"; do
  echo "cfg $cfg shape $shape"; ATOM_F6_CFG=$cfg timeout 120 $T/gemm_bench $shape 200 256 | grep -E "check|RESULT"; done; done > $O/bench_shapes.txt 2>&1
for c in 0 1001 1002 1004 1008 1003 1006 1007 1010 1014 1015 1064 1128 1192 1207; do echo -n "cfg $c "; ATOM_F6_CFG=$c timeout 60 $T/gemm_bench 4096 4096 4096 300 0 | grep RESULT; done > $O/abl.txt 2>&1
timeout 60 $T/trace_f6p > $O/trace.txt 2>&1
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  n=$(echo $set | cut -d' ' -f1)
  ATOM_F6_CFG=0 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$n -o g -- $T/gemm_bench 4096 4096 4096 40 0 > $O/pmc_$n.log 2>&1
done
python3 - <<PY > $O/pmc.txt
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_w4a4' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    print("  %-28s n=%d mean=%.5g"%(k,len(v),sum(v)/len(v)))
PY
rm -rf $O/pmc_*/
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo done
