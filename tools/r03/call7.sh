cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 600 python tools/cold_bench.py all > gpurun_out/r03/cold_bench.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03/cold_bench.txt | cut -c1-200
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench2.json 2> gpurun_out/r03/bench2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['frac'], d['abi_value'], d['cold'])
PY
