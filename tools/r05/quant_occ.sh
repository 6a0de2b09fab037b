# Round 5: how the quantisers' time depends on the number of resident workgroups per CU (unused LDS added to the launch)
mkdir -p gpurun_out/r05
O=gpurun_out/r05/quant_occupancy.txt
{
for extra in 0 1024 8192 16384 28672 57344; do
  echo "== extra LDS $extra B  (RMSNorm 25.6 KB + extra -> workgroups per CU by LDS: $((163840 / (25600 + extra))); reorder 17 KB + extra: $((163840 / (17408 + extra))))"
  ATOM_Q_EXTRA_LDS=$extra timeout 120 build/tools/quant_bench 4096 4096 200 2>&1 | grep -E "dequant_out=0" | grep -E "reorder|rmsnorm"
done
} > $O 2>&1
cat $O
