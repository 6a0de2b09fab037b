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

KERNELS = {"f6": "gemm_w4a4_f6q_kernel", "packed": "Cfg<256, 128, 3, 4, false", "wide": "Cfg<256, 256, 3, 4, true"}
CONTROLS = ("stream_lds_dma", "stream_vgpr", "stream_store")       # tools/probes/fetch_control.cpp: 512 MiB per launch each
CONTROL_KIB = 512 * 1024
ALGO_BYTES = 51380224       # SURVEY 8(d), M=N=K=4096


def kernel_source_sha():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.kernel_source_sha()


def main(raw, out):
    os.makedirs(out, exist_ok=True)
    bstats = glob.glob(os.path.join(raw, "block", "**", "*kernel_stats.csv"), recursive=True)
    if bstats:                                             # the decoder block (tools/block_bench.py) -> block_kernel_stats.csv
        rows = list(csv.DictReader(open(bstats[0])))
        with open(os.path.join(out, "block_kernel_stats.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
            for r in rows:
                w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
        if os.path.exists(os.path.join(raw, "block.log")):
            with open(os.path.join(out, "block_bench_under_rocprof.txt"), "w") as f:
                f.write(open(os.path.join(raw, "block.log")).read()[-4000:])
    stats = glob.glob(os.path.join(raw, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(out, "bench_kernel_stats.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    acc = {k: collections.defaultdict(list) for k in KERNELS}
    ctl = {k: collections.defaultdict(list) for k in CONTROLS}
    for f in glob.glob(os.path.join(raw, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for fmt, pat in KERNELS.items():
                if pat in r["Kernel_Name"]:
                    acc[fmt][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for name in CONTROLS:
                if name in r["Kernel_Name"]:
                    ctl[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # measured corrections: counter (KiB) per true KiB for each access type; fall back to the guide's figures when the control did not run
    corr = {"fetch_lds_dma": None, "fetch_vgpr": None, "write_store": None}
    if ctl["stream_lds_dma"].get("FETCH_SIZE"):
        corr["fetch_lds_dma"] = sum(ctl["stream_lds_dma"]["FETCH_SIZE"]) / len(ctl["stream_lds_dma"]["FETCH_SIZE"]) / CONTROL_KIB
    if ctl["stream_vgpr"].get("FETCH_SIZE"):
        corr["fetch_vgpr"] = sum(ctl["stream_vgpr"]["FETCH_SIZE"]) / len(ctl["stream_vgpr"]["FETCH_SIZE"]) / CONTROL_KIB
    if ctl["stream_store"].get("WRITE_SIZE"):
        corr["write_store"] = sum(ctl["stream_store"]["WRITE_SIZE"]) / len(ctl["stream_store"]["WRITE_SIZE"]) / CONTROL_KIB
    json.dump({"what": "rocprofv3 counter (KiB) per KiB actually streamed, tools/probes/fetch_control.cpp (512 MiB per launch), same kind of pass",
               "FETCH_SIZE_per_KiB_lds_dma": corr["fetch_lds_dma"], "FETCH_SIZE_per_KiB_vgpr_16B": corr["fetch_vgpr"],
               "WRITE_SIZE_per_KiB_store_16B": corr["write_store"],
               "samples": {k: {c: len(v) for c, v in d.items()} for k, d in ctl.items()}},
              open(os.path.join(out, "bench_traffic_controls.json"), "w"), indent=1)
    f_fetch = 1.0 / corr["fetch_lds_dma"] if corr["fetch_lds_dma"] else 2.0
    f_write = 1.0 / corr["write_store"] if corr["write_store"] else 1.0
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
                "correction": "FETCH_SIZE and WRITE_SIZE divided by what the same counters report per KiB of a known stream of the same "
                              "access type in the same kind of pass (bench_traffic_controls.json: LDS-DMA reads, 16-byte stores); without "
                              "the control: FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM section), WRITE_SIZE as is",
                "fetch_factor": f_fetch, "write_factor": f_write,
                "traffic_bytes_per_launch": int(round(f_fetch * fetch_kib * 1024 + f_write * write_kib * 1024)),
                "f6_operand_and_output_bytes_per_launch": 31 * 8192 * 104 + 31 * 4096 * 4 + 2 * 4096 * 128 + 2 * 4096 * 4096,   # records of both operands, float32 weight scales, keeper, fp16 output
                "kernel_source_sha256": kernel_source_sha(),
                "algorithmic_bytes_per_launch": ALGO_BYTES}, open(os.path.join(out, f"bench_hbm_traffic_{fmt}.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
