build/gemm_bench 4096 4096 4096 30 0 | grep RESULT
build/gemm_bench 4096 4096 4096 300 0 | grep RESULT
python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench.py', d['value'], d['roofline']['kernel_us'])"
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench.py 1000', d['value'], d['roofline']['kernel_us'])"
build/gemm_bench 4096 4096 4096 30 0 | grep RESULT
