# GEMM sweeps of BASELINE.md 1a / 1b (N=K=4096, M sweep) and config 5 (Llama-13B shapes); run on the GPU box.
R=$GRAFT_REPO_ROOT
export ATOM_WS=1
echo "# M sweep, N=K=4096 (reference RTX 4090 numbers: BASELINE.md 1a)"
for M in 1 8 16 32 64 128 256 512 1024 2048 4096; do
  $R/build/tools/gemm_bench $M 4096 4096 200 0 | grep RESULT
done
echo "# config 5: Llama-13B shapes"
for NK in "5120 5120" "13824 5120" "5120 13824"; do
  for M in 1 16 64 256 2048; do
    $R/build/tools/gemm_bench $M $NK 100 0 | grep RESULT
  done
done
