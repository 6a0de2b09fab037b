cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02/bench_line.json 2> gpurun_out/r02/bench_err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02/bench_line.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','abi_value')}, d['roofline']['frac'], d['roofline']['kernel_us'])
print({k:(v['value'],v['kernel_us'],v['bit_identical_to_headline_output']) for k,v in d['other_operand_formats'].items()})
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
