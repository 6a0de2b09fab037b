export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$R/build/tools; O=$R/gpurun_out/c19; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_quant.py tests/test_gpu_ref.py tests/test_gpu_e2e.py tests/test_gpu_block.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest.txt
for M in 1 16 256 4096 65536; do echo "# M=$M H=4096 packed"; $T/quant_bench $M 4096 100 | grep -E "rmsnorm" ; done > $O/quant.txt 2>&1
python tools/fused_bench.py 2>&1 | grep FRESULT >> $O/quant.txt
cat $O/pytest.txt $O/quant.txt
