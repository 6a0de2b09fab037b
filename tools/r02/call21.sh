cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
{ for c in 1005 1009; do for M in 256 1024; do timeout 120 build/tools/trace_f6kg $c $M 4096 4096; done; done; } > gpurun_out/r02/trace_kg.txt 2>&1
head -150 gpurun_out/r02/trace_kg.txt
