cd $GRAFT_REPO_ROOT
timeout 600 python tools/r02/gateup_bench.py 2>&1 | grep -vE "amdgpu.ids" | tail -8
