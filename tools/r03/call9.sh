cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_gateup.py tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | tail -5
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench3.json 2> gpurun_out/r03/bench3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench3.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['frac'], d['abi_value'], d['cold']['kernel_us'], {k:v['kernel_us'] for k,v in d['other_operand_formats'].items()})
PY
