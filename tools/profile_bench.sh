# Round profile of the headline command (run on the GPU box through gpurun; raw outputs in /tmp on the box, summaries
# written by tools/summarize_profile.py into gpurun_out/prof_summary/ -- copy those into profiles/rNN/).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/atom_prof_raw          # (raw rocprofv3 output stays on the box: gpurun copies at most 64 MiB of gpurun_out/ back)
rm -rf $OUT; mkdir -p $OUT
CMD="python $R/bench.py --no-cpu-baseline --no-configs --no-block --no-decode --steps 500 --warmup 100"   # 500 timed + 100 warm-up launches per operand format (+ the 1500-launch ramp)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- $CMD > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- $CMD > $OUT/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -o bench -- $CMD > $OUT/sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM --output-format csv -d $OUT/sq2 -o bench -- $CMD > $OUT/sq2.log 2>&1
tail -1 $OUT/stats.log
# controls for the FETCH_SIZE / WRITE_SIZE corrections (build/tools/fetch_control <- `make -C atom_amd/csrc tools probes`; tools/probes/fetch_control.cpp: 512 MiB streamed once per launch by LDS-DMA, by
# 16-byte loads into registers, by 16-byte stores), in the same kind of pass
CTL="$R/build/tools/fetch_control 512 5"
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/ctl_fetch -o ctl -- $CTL > $OUT/ctl_fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/ctl_write -o ctl -- $CTL > $OUT/ctl_write.log 2>&1
# config 4 (the Llama-7B decoder block of the default bench line's `block` key) under the same tracer: per-kernel split of one block
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/block -o block -- python $R/tools/block_bench.py > $OUT/block.log 2>&1
tail -12 $OUT/block.log
cd $R && python tools/summarize_profile.py $OUT $R/gpurun_out/prof_summary
