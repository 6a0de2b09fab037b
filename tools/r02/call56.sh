cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['roofline']['kernel_us'], {k:v['kernel_us'] for k,v in d['other_operand_formats'].items()})"
done
ATOM_F6=1 timeout 100 build/tools/gemm_bench 4096 4096 4096 300 0 | grep RESULT | cut -c1-120
timeout 100 build/tools/gemm_bench 4096 4096 4096 200 0 | grep RESULT | cut -c1-120
