# rocprofv3 evidence for the decode GEMMs (M = 1: gemv1 kernel; M = 16: decode-batch kernel), BASELINE config 2 and the Llama shapes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$R/build/tools; O=$R/gpurun_out/prof_decode; rm -rf $O; mkdir -p $O
cd /tmp
for shape in "1 4096 4096" "16 4096 4096" "1 13824 5120" "1 11008 4096"; do
  tag=$(echo $shape | tr ' ' 'x')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -o g -- $T/gemm_bench $shape 200 0 > $O/stats_$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$tag -o g -- $T/gemm_bench $shape 50 0 > $O/fetch_$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$tag -o g -- $T/gemm_bench $shape 50 0 > $O/write_$tag.log 2>&1
done
python3 - <<PY > $R/gpurun_out/prof_decode_summary.txt
import csv, glob, os, collections
O = "$O"
print("shape (M N K) | kernel | calls | avg us (rocprofv3 --stats) | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | algorithmic KiB")
for d in sorted(glob.glob(O + "/stats_*/")):
    tag = os.path.basename(d.rstrip("/"))[6:]
    M, N, K = map(int, tag.split("x"))
    K4 = K - 128; G = K4 // 128
    alg = (M * K4 // 2 + N * K4 // 2 + 128 * (M + N) + 2 * (M * G + N * G + M + N) + 2 * M * N) / 1024
    st = glob.glob(d + "**/*kernel_stats.csv", recursive=True)
    rows = [r for r in csv.DictReader(open(st[0])) if "gemv" in r["Name"] or "skinny" in r["Name"] or "gemm_w4a4" in r["Name"]]
    pm = collections.defaultdict(list)
    for kind in ("fetch", "write"):
        for f in glob.glob(O + f"/{kind}_{tag}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if any(x in r["Kernel_Name"] for x in ("gemv", "skinny", "gemm_w4a4")):
                    pm[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for r in rows:
        name = r["Name"][:60]
        f = pm.get((name, "FETCH_SIZE"), [0]); w = pm.get((name, "WRITE_SIZE"), [0])
        print(f"{M} {N} {K} | {r['Name'][:70]} | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {sum(f)/len(f):.0f} | {sum(w)/len(w):.0f} | {alg:.0f}")
PY
cat $R/gpurun_out/prof_decode_summary.txt
