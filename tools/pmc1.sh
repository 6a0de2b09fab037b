export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc/p1 -o p1 -- $R/build/gemm_bench 4096 4096 4096 5 0 > $R/gpurun_out/pmc/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc/p2 -o p2 -- $R/build/gemm_bench 4096 4096 4096 5 0 > $R/gpurun_out/pmc/p2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc/st -o st -- $R/build/gemm_bench 4096 4096 4096 50 0 > $R/gpurun_out/pmc/st.log 2>&1
ls -R $R/gpurun_out/pmc | head -30
