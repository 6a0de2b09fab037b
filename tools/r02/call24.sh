cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python tools/r02/sweep.py > gpurun_out/r02/sweep.txt 2>&1; tail -40 gpurun_out/r02/sweep.txt
