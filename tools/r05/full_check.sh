#!/bin/bash
# Round 5: the whole GPU suite + the default driver line on the current tree
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/pytest_gpu_full.txt
cat $O/pytest_gpu_full.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.err
python3 - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_us"], d["roofline"]["traffic"])
for r in d["configs"]["rows"]:
    print({k:r[k] for k in ("M","N","K","hot_us","cold_us") }, {k:v for k,v in r.items() if k.startswith(("wcached","f6_us","f6_cold"))})
print(d.get("block",{}).get("block_ms"), d.get("block",{}).get("gemm_tops"))
PY
