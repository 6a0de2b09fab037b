export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c16; mkdir -p $O; cd $R
echo "== one-round-trip kernel, default row policy" > $O/decode.txt; ATOM_LIB=$R/build/tools/libatom_hip.so python tools/r02/decode_probe.py 2>&1 | grep -v amdgpu >> $O/decode.txt
echo "== ATOM_GEMV1_ROWS2=0" >> $O/decode.txt; ATOM_LIB=$R/build/tools/libatom_hip.so ATOM_GEMV1_ROWS2=0 python tools/r02/decode_probe.py 2>&1 | grep -v amdgpu >> $O/decode.txt
echo "== ATOM_GEMV1_ROWS2=1" >> $O/decode.txt; ATOM_LIB=$R/build/tools/libatom_hip.so ATOM_GEMV1_ROWS2=1 python tools/r02/decode_probe.py 2>&1 | grep -v amdgpu >> $O/decode.txt
timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x 2>&1 | tail -3 >> $O/decode.txt
cat $O/decode.txt
