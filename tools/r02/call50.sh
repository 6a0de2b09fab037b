cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 600 python tools/r02/sweep.py > gpurun_out/r02/sweep2.txt 2>&1; grep -vE "amdgpu.ids" gpurun_out/r02/sweep2.txt | tail -38 | cut -c1-120
