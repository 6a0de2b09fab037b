# Ablations of the F6 256x256 kernel (build/tools = library built with -DATOM_TOOLS: make -C atom_amd/csrc tools)
# mask bits: 1 no LDS-DMA after the prologue, 2 no de-quantisation, 4 no MFMA, 8 no activation-fragment refills
R=$GRAFT_REPO_ROOT
export ATOM_F6=1
for c in 0 101 102 104 108 103 106 110 107 114 115; do
  echo -n "ATOM_F6_CFG=$c  "; ATOM_F6_CFG=$c $R/build/tools/gemm_bench 4096 4096 4096 300 0 | grep RESULT
done
