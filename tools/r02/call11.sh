export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c11; mkdir -p $O; cd $R
python tools/r02/sweep.py > $O/sweep.txt 2> $O/sweep.err
tail -3 $O/sweep.err; cat $O/sweep.txt
