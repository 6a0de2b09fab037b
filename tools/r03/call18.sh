cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python tools/sweep.py > gpurun_out/r03/sweep.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03/sweep.txt | cut -c1-150
