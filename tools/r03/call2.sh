cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_gpu_block.py -m gpu -q -x -s -k "7b_block" > gpurun_out/r03/block7b.txt 2>&1
grep -E "stage-wise|vs the reference golden|code flips|passed|failed|^E " gpurun_out/r03/block7b.txt | cut -c1-1800
