# F6 tile geometries vs the INT8 kernels over prefill shapes (us, avg of 200 launches); run on the GPU box.
R=$GRAFT_REPO_ROOT
t() { "$@" | grep RESULT | sed 's/.*avg \([0-9.]*\) us.*/\1/'; }
printf "%-18s %8s %8s %8s %8s %8s %8s\n" "M N K" f6_256 f6_256x128 f6_128 f6_64sk int8wide int8
for NK in "4096 4096" "11008 4096" "4096 11008"; do
  for M in 256 512 1024 1536 2048 3072 4096 8192; do
    a=$(ATOM_F6=1 ATOM_F6_CFG=0 t $R/build/tools/gemm_bench $M $NK 200 0)
    b=$(ATOM_F6=1 ATOM_F6_CFG=1 t $R/build/tools/gemm_bench $M $NK 200 0)
    c=$(ATOM_F6=1 ATOM_F6_CFG=3 t $R/build/tools/gemm_bench $M $NK 200 0)
    d=$(ATOM_F6=1 ATOM_F6_CFG=2 ATOM_WS=1 t $R/build/tools/gemm_bench $M $NK 200 0)
    e=$(ATOM_AWIDE=1 ATOM_WS=1 t $R/build/tools/gemm_bench $M $NK 200 0)
    f=$(ATOM_WS=1 t $R/build/tools/gemm_bench $M $NK 200 0)
    printf "%-18s %8s %8s %8s %8s %8s %8s\n" "$M $NK" $a $b $c $d $e $f
  done
done
