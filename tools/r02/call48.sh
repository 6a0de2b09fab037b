cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python tools/r02/sweep.py > gpurun_out/r02/sweep2.txt 2>&1; grep -vE "amdgpu.ids" gpurun_out/r02/sweep2.txt | tail -38
