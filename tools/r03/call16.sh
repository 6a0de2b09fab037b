cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | cut -c1-300
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench4.json 2> gpurun_out/r03/bench4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench4.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['frac'], d['abi_value'], d['cold']['kernel_us'], {k:(v['kernel_us'], v['bit_identical_to_headline_output']) for k,v in d['other_operand_formats'].items()})
PY
