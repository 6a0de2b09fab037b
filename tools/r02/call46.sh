cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python tools/decode_step_bench.py > gpurun_out/r02/decode_step.txt 2>&1; tail -14 gpurun_out/r02/decode_step.txt | cut -c1-180
