export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$R/build/tools; O=$R/gpurun_out/c6; mkdir -p $O; cd $R
export ATOM_F6=1
for d in rand zero const small rand; do for c in 0 1002; do echo -n "data $d cfg $c "; env $( [ $d = rand ] || echo ATOM_DATA=$d ) ATOM_F6_CFG=$c timeout 60 $T/gemm_bench 4096 4096 4096 300 0 | grep RESULT; done; done > $O/data.txt 2>&1
cat $O/data.txt
