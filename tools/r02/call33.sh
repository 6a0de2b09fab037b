cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
{ for fmt in 0 512; do for M in 4096 65536; do echo "## fmt=$fmt M=$M"; ATOM_QB_FMT=$fmt timeout 120 build/tools/quant_bench $M 4096 $([ $M = 4096 ] && echo 200 || echo 20); done; done; } > gpurun_out/r02/quant_bench3.txt 2>&1
grep -E "^##|dequant_out=0" gpurun_out/r02/quant_bench3.txt | grep -v silu | cut -c1-200
timeout 1200 python -m pytest tests/test_gpu_quant.py tests/test_gpu_ref.py tests/test_gpu_block.py tests/test_gpu_e2e.py -q 2>&1 | tail -5
