# every F6 tile geometry on a grid of shapes (tools build: ATOM_F6_CFG forces the geometry; "d" = the dispatch's own pick)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for nk in "4096 4096" "11008 4096" "4096 11008" "5120 5120" "13824 5120" "5120 13824"; do
  for m in 256 384 512 768 1024 1536 2048 3072; do
    for cfg in d 0 3 6 8 9 12; do
      if [ $cfg = d ]; then e=""; else e="ATOM_F6_CFG=$cfg"; fi
      env $e ATOM_F6=1 timeout 60 build/tools/gemm_bench $m $nk 60 0 2>&1 | grep -E "RESULT" | sed "s/^/cfg=$cfg /" | awk '{print $1, $4, $5, $6, $8}'
    done
  done
done 2>&1 | tee gpurun_out/r03/cfg_sweep.txt | tail -5
