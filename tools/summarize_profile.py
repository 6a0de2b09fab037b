"""Condense the rocprofv3 outputs of tools/profile_bench.sh into the small files committed under profiles/rNN/:
   bench_kernel_stats.csv        per-kernel call count / average duration (rocprofv3 --stats)
   bench_pmc_summary_<fmt>.json  mean counter values per launch of the three GEMM kernels
   bench_hbm_traffic_<fmt>.json  HBM bytes per launch (FETCH_SIZE x 2 on gfx950 wide reads + WRITE_SIZE; MI355X_MICROARCH.md)
usage: python tools/summarize_profile.py <raw dir> <out dir>"""
import collections
import csv
import glob
import json
import os
import sys

KERNELS = {"f6": "gemm_w4a4_f6q_kernel", "packed": "Cfg<256, 128, 3, 4, false>", "wide": "Cfg<256, 256, 3, 4, true>"}
ALGO_BYTES = 51380224       # SURVEY 8(d), M=N=K=4096


def kernel_source_sha():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.kernel_source_sha()


def main(raw, out):
    os.makedirs(out, exist_ok=True)
    stats = glob.glob(os.path.join(raw, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(out, "bench_kernel_stats.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    acc = {k: collections.defaultdict(list) for k in KERNELS}
    for f in glob.glob(os.path.join(raw, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for fmt, pat in KERNELS.items():
                if pat in r["Kernel_Name"]:
                    acc[fmt][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for fmt, counters in acc.items():
        if not counters:
            continue
        summ = {k: {"n": len(v), "mean": sum(v) / len(v)} for k, v in sorted(counters.items())}
        json.dump(summ, open(os.path.join(out, f"bench_pmc_summary_{fmt}.json"), "w"), indent=1)
        if "FETCH_SIZE" in summ and "WRITE_SIZE" in summ:
            fetch_kib, write_kib = summ["FETCH_SIZE"]["mean"], summ["WRITE_SIZE"]["mean"]
            json.dump({
                "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --no-cpu-baseline",
                "kernel": KERNELS[fmt] + "  M=N=K=4096", "operand_format": fmt,
                "FETCH_SIZE_KiB_per_launch": fetch_kib, "WRITE_SIZE_KiB_per_launch": write_kib,
                "correction": "gfx950 rocprofv3 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads "
                              "(MI355X_MICROARCH.md, HBM section): doubled; WRITE_SIZE taken as is",
                "traffic_bytes_per_launch": int(round(2 * fetch_kib * 1024 + write_kib * 1024)),
                "kernel_source_sha256": kernel_source_sha(),
                "algorithmic_bytes_per_launch": ALGO_BYTES}, open(os.path.join(out, f"bench_hbm_traffic_{fmt}.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
