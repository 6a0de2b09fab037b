cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_quant.py tests/test_gpu_gateup.py tests/test_gpu_ref.py tests/test_gpu_e2e.py tests/test_gpu_block.py tests/test_gpu_pack.py -q 2>&1 | tail -3
timeout 120 python tools/r02/dbg_f6codes.py 2>&1 | grep -c ok
ATOM_QB_FMT=512 timeout 100 build/tools/quant_bench 65536 4096 20 | grep "dequant_out=0" | cut -c1-120
