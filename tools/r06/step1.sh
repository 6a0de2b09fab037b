#!/bin/bash
# Round 6, step 1 (one gpurun call): the GPU suite on the hardened C ABI (ATOM_WS_VERIFY, atom_check_scale_pairs, cached weight never
# overwritten), then same-box A/Bs of the headline at 4096^3 in the BF6 format:
#   new   = working tree (token scale read as ds_read_b64: no bank conflict)      build/tools/gemm_bench
#   sa32  = the round-5 read (ds_read_b32, 2-way conflict; -DATOM_SA_B32)         build/ab/sa32/gemm_bench
#   x16w8 / x16w16 = the SAME first-generation micro-tile kernel body as 8 waves of 64 x 128 (cfg 41) and as 16 waves of 64 x 64
#           (cfg 42: four waves per SIMD, 128 registers -- the compiler spills 94 of them)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_step1.txt 2>&1; echo "pytest rc $?" >> $O/pytest_step1.txt
tail -5 $O/pytest_step1.txt
{
  for r in 1 2 3; do
    echo "== new f6 4096^3 (round $r)"; ATOM_F6=1 build/tools/gemm_bench 4096 4096 4096 400 $((r==1?64:0)) | grep -E "RESULT|check"
    echo "== sa32 f6 4096^3 (round $r)"; ATOM_F6=1 build/ab/sa32/gemm_bench 4096 4096 4096 400 0 | grep RESULT
  done
  for r in 1 2; do
    echo "== x16 8 waves (cfg 41) 4096^3 (round $r)"; ATOM_F6=1 ATOM_F6_CFG=41 build/tools/gemm_bench 4096 4096 4096 300 64 | grep -E "RESULT|check"
    echo "== x16 16 waves (cfg 42) 4096^3 (round $r)"; ATOM_F6=1 ATOM_F6_CFG=42 build/tools/gemm_bench 4096 4096 4096 300 64 | grep -E "RESULT|check"
  done
  echo "== new f6 8192^3"; ATOM_F6=1 build/tools/gemm_bench 8192 8192 8192 60 0 | grep RESULT
  echo "== sa32 f6 8192^3"; ATOM_F6=1 build/ab/sa32/gemm_bench 8192 8192 8192 60 0 | grep RESULT
} > $O/ab_headline.txt 2>&1
cat $O/ab_headline.txt
bash tools/pmc_gemm.sh f6new 4096 4096 4096 ATOM_F6=1 > $O/pmc_f6_new.txt 2>&1; grep -E "LDS|stats|WAIT|MFMA" $O/pmc_f6_new.txt
