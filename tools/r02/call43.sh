cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 300 python tools/r02/quant_latency_probe.py > gpurun_out/r02/quant_latency2.txt 2>&1; grep -E "tiny|H=" gpurun_out/r02/quant_latency2.txt
timeout 900 python -m pytest tests/test_gpu_quant.py tests/test_gpu_e2e.py tests/test_gpu_ref.py -q 2>&1 | tail -3
