cd $GRAFT_REPO_ROOT
S=$(date +%s)
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['roofline']['kernel_us'], d['abi_value'], {k:(v['kernel_us'],v['warmup'],v['steps']) for k,v in d['other_operand_formats'].items()})"
echo "wall $(( $(date +%s) - S )) s"
