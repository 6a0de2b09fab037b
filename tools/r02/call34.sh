cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 300 python tools/r02/decode_probe.py > gpurun_out/r02/decode_probe2.txt 2>&1; grep -E "M=1|tiny" gpurun_out/r02/decode_probe2.txt | head -12
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -k "decode or gemv or vs_oracle or c_contract" 2>&1 | tail -4
