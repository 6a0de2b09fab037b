cd $GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -12
timeout 600 python tools/r02/sweep.py > gpurun_out/r02/sweep.txt 2>&1; tail -40 gpurun_out/r02/sweep.txt
