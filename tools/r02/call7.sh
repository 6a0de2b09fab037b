export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$R/build/tools; O=$R/gpurun_out/c7; mkdir -p $O; cd $R
export ATOM_F6=1
for c in 0 3048 5096 0 3048 5096; do echo -n "cfg $c "; ATOM_F6_CFG=$c timeout 60 $T/gemm_bench 4096 4096 4096 300 0 | grep RESULT; done > $O/abl.txt 2>&1
cat $O/abl.txt
