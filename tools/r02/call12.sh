export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$R/build/tools; O=$R/gpurun_out/c12; mkdir -p $O; cd $R
export ATOM_F6=1 ATOM_WS=1
t() { "$@" | grep RESULT | sed 's/.*avg \([0-9.]*\) us.*/\1/'; }
echo "# F6 mid-M geometries, avg us: cfg3 = 128x128 4 waves, cfg4 = 128x128 8 waves (s1/s2 = K splits), cfg2 = 64x128 split-K, cfg0 = 256x256" > $O/midm.txt
for NK in "4096 4096" "11008 4096" "4096 11008" "5120 5120" "13824 5120"; do for M in 256 512 768 1024 1536 2048 3072; do
  a=$(ATOM_F6_CFG=3 ATOM_F6_SPLITS3=1 t $T/gemm_bench $M $NK 100 0); b=$(ATOM_F6_CFG=3 ATOM_F6_SPLITS3=2 t $T/gemm_bench $M $NK 100 0)
  c=$(ATOM_F6_CFG=4 ATOM_F6_SPLITS3=1 t $T/gemm_bench $M $NK 100 0); d=$(ATOM_F6_CFG=4 ATOM_F6_SPLITS3=2 t $T/gemm_bench $M $NK 100 0)
  e=$(ATOM_F6_CFG=2 t $T/gemm_bench $M $NK 100 0); f=$(ATOM_F6_CFG=0 t $T/gemm_bench $M $NK 100 0); g=$(t $T/gemm_bench $M $NK 100 0)
  echo "M=$M N,K=$NK | cfg3 s1 $a s2 $b | cfg4 s1 $c s2 $d | cfg2 $e | cfg0 $f | dispatch $g"; done; done >> $O/midm.txt 2>&1
ATOM_F6_CFG=4 $T/gemm_bench 1000 1408 640 20 1000 | grep -E "check|RESULT" >> $O/midm.txt
ATOM_F6_CFG=4 ATOM_F6_SPLITS3=2 $T/gemm_bench 777 4096 1152 20 777 | grep -E "check|RESULT" >> $O/midm.txt
cat $O/midm.txt
