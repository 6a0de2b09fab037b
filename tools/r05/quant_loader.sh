# Round 5: the activation quantisers with the loader wave (quant_kernels.hip) -- parity, then timing at the ring depths 2..5.
mkdir -p gpurun_out/r05
O=gpurun_out/r05/quant_loader.txt
{
timeout 900 python -m pytest tests/test_gpu_quant.py tests/test_gpu_ref.py -m gpu -x -q 2>&1 | tail -5
for shape in "1024 4096" "4096 4096" "16384 4096" "65536 4096" "4096 8192" "4096 5120"; do
  for nst in default 2 3 4 5; do
    if [ $nst = default ]; then unset ATOM_Q_NST; else export ATOM_Q_NST=$nst; fi
    echo "== M H = $shape  nst=$nst"
    timeout 120 build/tools/quant_bench $shape 200 2>&1 | grep -E "sim|kernel" | head -12
  done
done
unset ATOM_Q_NST
echo "== BF6 records, default depth"
ATOM_QB_FMT=512 timeout 120 build/tools/quant_bench 4096 4096 200 2>&1 | head -14
ATOM_QB_FMT=512 timeout 120 build/tools/quant_bench 65536 4096 50 2>&1 | head -14
} > $O 2>&1
tail -150 $O
