# non-temporal weight loads in the decode kernels: working tree vs HEAD (build/ab/base), same box; decode GEMMs hot / cold, then the layer
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
cp atom_amd/libatom_hip.so /tmp/new.so
cat > /tmp/cg.py <<'PY'
import sys
sys.path.insert(0, "tools")
import cold_bench as c
for (M, N, K) in [(1, 4096, 4096), (2, 4096, 4096), (4, 4096, 4096), (8, 4096, 4096), (4, 11008, 4096), (8, 11008, 4096), (8, 4096, 11008), (16, 11008, 4096)]:
    c.gemm_row(M, N, K)
c.layer_main(batches=(1, 2, 4, 8, 16))
PY
for rep in 1 2; do
  for v in new base; do
    if [ $v = new ]; then cp /tmp/new.so atom_amd/libatom_hip.so; else cp build/ab/base/libatom_hip.so atom_amd/libatom_hip.so; fi
    timeout 400 python /tmp/cg.py 2>&1 | grep -E "hot" | sed "s/^/$v rep$rep: /" | cut -c1-150
  done
done 2>&1 | tee gpurun_out/r03/nt_ab.txt
cp /tmp/new.so atom_amd/libatom_hip.so
