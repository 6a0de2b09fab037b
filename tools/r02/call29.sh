cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 300 build/tools/issue_probe bf6u16 > gpurun_out/r02/issue_probe_pk.txt 2>&1
timeout 300 build/tools/issue_probe bf6s16 2>&1 | grep -E "VALU_only|PK_only" >> gpurun_out/r02/issue_probe_pk.txt
grep -E "deq_pk|t\+2\),deq\(" gpurun_out/r02/issue_probe_pk.txt
