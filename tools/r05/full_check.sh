#!/bin/bash
# Round 5: the whole GPU suite + the driver's bench command on the current tree (one gpurun call)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_full.txt 2>&1; echo "pytest rc $?" >> $O/pytest_full.txt
tail -12 $O/pytest_full.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r05/bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "kernel_us", d["roofline"]["kernel_us"], "frac", d["roofline"]["frac"], "cold", d.get("cold",{}).get("kernel_us"))
print("block", json.dumps(d.get("block"))[:600])
for r in d.get("configs",{}).get("rows",[]):
    print(r["config"], r["M"], r["N"], r["K"], "hot", r["hot_us"], "cold", r["cold_us"], "f6", r.get("f6_us"))
for k,v in d.get("other_operand_formats",{}).items(): print(k, v["kernel_us"], v["bit_identical_to_headline_output"])
P
tail -5 $O/bench_err.txt
