cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python tools/r02/packed_route_probe.py 2>&1 | grep "M="
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o pr -- python $GRAFT_REPO_ROOT/tools/r02/packed_route_probe.py 1024 > /dev/null 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
