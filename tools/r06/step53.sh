#!/bin/bash
# Round 6, step 53: where the kernel arguments live (HIP_FORCE_DEV_KERNARG: the runtime's kernarg pool in device memory instead of host memory)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
{
for v in 0 1 0 1; do echo "-- HIP_FORCE_DEV_KERNARG=$v"; HIP_FORCE_DEV_KERNARG=$v timeout 120 build/tools/kp0; done
for v in 0 1 0 1; do echo "== HIP_FORCE_DEV_KERNARG=$v"; HIP_FORCE_DEV_KERNARG=$v timeout 300 python tools/cold_bench.py layer 1,16 2>&1 | grep "^batch"; done
} 2>&1 | tee $O/ab_dev_kernarg.txt
