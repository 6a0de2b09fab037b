cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
{ for rb in 1 4 8 16; do for fmt in 0 512; do for M in 4096 16384 65536; do echo "## rb=$rb fmt=$fmt M=$M"; ATOM_QUANT_RB=$rb ATOM_QB_FMT=$fmt timeout 120 build/tools/quant_bench $M 4096 $([ $M = 65536 ] && echo 20 || echo 100) | grep "dequant_out=0" | grep -v silu | grep "mode=sim"; done; done; done; } > gpurun_out/r02/quant_rb.txt 2>&1
cat gpurun_out/r02/quant_rb.txt | cut -c1-150
