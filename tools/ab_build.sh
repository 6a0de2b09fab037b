#!/bin/bash
# A/B builds for same-box comparisons: tools/ab_build.sh <name> [git-rev]
#   builds build/ab/<name>/libatom_hip.so (-DATOM_TOOLS) + gemm_bench from the kernel sources of <git-rev> (default: the working tree),
#   so that one gpurun call can time two versions of a kernel on the SAME box (different boxes of the pool differ by +-3 %).
set -e
cd "$(dirname "$0")/.."
name=$1; rev=$2
out=build/ab/$name; src=$out/src
rm -rf $out; mkdir -p $src/atom_amd/csrc $src/include $src/tools
if [ -n "$rev" ]; then
  git archive $rev atom_amd/csrc include tools/gemm_bench.cpp tools/quant_bench.cpp | tar -x -C $src
else
  cp atom_amd/csrc/*.hip atom_amd/csrc/*.h $src/atom_amd/csrc/; cp include/*.h $src/include/; cp tools/gemm_bench.cpp tools/quant_bench.cpp $src/tools/
fi
objs=""
for f in $src/atom_amd/csrc/*.hip; do
  o=$out/$(basename $f .hip).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -DATOM_TOOLS $ABFLAGS -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $out/libatom_hip.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $src/tools/gemm_bench.cpp -o $out/gemm_bench -L$out -latom_hip -Wl,-rpath,'$ORIGIN' 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $src/tools/quant_bench.cpp -o $out/quant_bench -L$out -latom_hip -Wl,-rpath,'$ORIGIN' 2>/dev/null || true
rm -f $objs; rm -rf $src
ls -la $out
