#!/bin/bash
# Round 6, step 55: decode attention where the pairs fill the chip -- 8-wave workgroups at two waves per SIMD (fewer, longer KV splits: fewer
# wave prologues per pair) against the shipped 12-wave workgroups at three
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
run() { echo "== $*"; env "$@" timeout 300 python tools/cold_bench.py layer $B 2>&1 | grep "^batch"; }
{
B=16
run ATOM_LIB=$PWD/build/tools/libatom_hip.so
run ATOM_LIB=$PWD/build/ab/wgm8/libatom_hip.so ATOM_DECODE_SPLITS=4
run ATOM_LIB=$PWD/build/ab/wgm8/libatom_hip.so ATOM_DECODE_SPLITS=2
run ATOM_LIB=$PWD/build/tools/libatom_hip.so ATOM_DECODE_SPLITS=4
run ATOM_LIB=$PWD/build/tools/libatom_hip.so ATOM_DECODE_SPLITS=3
B=8
run ATOM_LIB=$PWD/build/tools/libatom_hip.so
run ATOM_LIB=$PWD/build/ab/wgm8/libatom_hip.so ATOM_DECODE_SPLITS=8
run ATOM_LIB=$PWD/build/ab/wgm8/libatom_hip.so ATOM_DECODE_SPLITS=4
B=32
run ATOM_LIB=$PWD/build/tools/libatom_hip.so
run ATOM_LIB=$PWD/build/ab/wgm8/libatom_hip.so ATOM_DECODE_SPLITS=2
B=64
run ATOM_LIB=$PWD/build/tools/libatom_hip.so
run ATOM_LIB=$PWD/build/ab/wgm8/libatom_hip.so ATOM_DECODE_SPLITS=2
run ATOM_LIB=$PWD/build/ab/wgm8/libatom_hip.so ATOM_DECODE_SPLITS=1
} 2>&1 | tee $O/ab_decode_wgm8.txt
