export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$R/build/tools; O=$R/gpurun_out/c4; mkdir -p $O; cd $R
export ATOM_F6=1
for c in 0 1256 1257 1512 1513 1768 1769 1008 1009 1520 1001; do echo -n "cfg $c "; ATOM_F6_CFG=$c timeout 60 $T/gemm_bench 4096 4096 4096 300 0 | grep RESULT; done > $O/abl.txt 2>&1
cat $O/abl.txt
