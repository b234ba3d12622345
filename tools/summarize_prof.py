#!/usr/bin/env python
"""Turn a gpurun_out/prof/ capture (rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE,
--pmc WRITE_SIZE; three separate passes of the same bench.py command) into the committed
summaries under profiles/.  Usage: tools/summarize_prof.py <tag>   (e.g. r01_v1)

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE under-reports *wide coalesced streaming* reads by exactly 2x, so streaming kernels
(NTT passes, vector ops) get fetch x2; gather-style kernels (MSM bucket accumulation reads
64-byte points at random) are left uncorrected.  Both raw and corrected values are recorded.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
STREAMING = ("k_ntt_pass", "k_ntt_last", "k_vec_op", "k_scale", "k_distribute_powers", "k_msm_lds_sweep", "k_msm_digits")


def short(name):
    return name.split("(")[0].replace("zk::", "").replace("void ", "")


def pmc(path):
    d = collections.defaultdict(list)
    if not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return d


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    shutil.copy(os.path.join(SRC, "kt", "kt_kernel_stats.csv"), os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
    stats = {short(r["Name"]): r for r in csv.DictReader(open(os.path.join(SRC, "kt", "kt_kernel_stats.csv")))}
    fetch, write = pmc(os.path.join(SRC, "fetch", "fetch_counter_collection.csv")), pmc(os.path.join(SRC, "write", "write_counter_collection.csv"))
    rows = []
    for k, r in stats.items():
        f = sum(fetch[k]) / len(fetch[k]) * 1024 if fetch.get(k) else None
        w = sum(write[k]) / len(write[k]) * 1024 if write.get(k) else None
        corr = 2.0 if any(k.startswith(s) for s in STREAMING) else 1.0
        rows.append({
            "kernel": k, "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"]),
            "fetch_bytes_raw": f, "fetch_correction": corr, "write_bytes": w,
            "hbm_bytes_per_launch": (f * corr if f is not None else 0) + (w or 0) if (f is not None or w is not None) else None,
        })
    rows.sort(key=lambda x: -x["pct"])
    json.dump(rows, open(os.path.join(out_dir, f"{tag}_summary.json"), "w"), indent=1)
    with open(os.path.join(out_dir, f"{tag}_summary.md"), "w") as fo:
        fo.write(f"# rocprofv3 summary `{tag}` -- bench.py (MSM 2^20 + NTT 2^20 per step), MI355X\n\n")
        fo.write("Passes: `--kernel-trace --stats`; `--pmc FETCH_SIZE`; `--pmc WRITE_SIZE` (separate runs).\n\n")
        fo.write("| kernel | calls | avg us | % time | FETCH raw MiB | corr | WRITE MiB | HBM MiB/launch |\n|---|---|---|---|---|---|---|---|\n")
        for x in rows:
            mib = lambda v: "-" if v is None else f"{v / 2**20:.1f}"
            fo.write(f"| {x['kernel']} | {x['calls']} | {x['avg_us']:.1f} | {x['pct']:.2f} | {mib(x['fetch_bytes_raw'])} | x{x['fetch_correction']:.0f} | {mib(x['write_bytes'])} | {mib(x['hbm_bytes_per_launch'])} |\n")
        bench_log = os.path.join(SRC, "bench_kt.log")
        if os.path.exists(bench_log):
            for line in open(bench_log):
                if line.startswith("{"):
                    fo.write("\nbench line under the profiler:\n\n```json\n" + line.strip() + "\n```\n")
    b = next((x for x in rows if x["kernel"] == "k_msm_buckets"), None)
    if b and b["hbm_bytes_per_launch"]:
        json.dump({"msm_buckets_bytes_per_launch": b["hbm_bytes_per_launch"], "source": f"profiles/{tag}_summary.json",
                   "note": "FETCH_SIZE (KiB, gather reads: uncorrected) + WRITE_SIZE (KiB) per launch of k_msm_buckets"},
                  open(os.path.join(out_dir, "traffic_r01.json"), "w"), indent=1)
    print(open(os.path.join(out_dir, f"{tag}_summary.md")).read())


if __name__ == "__main__":
    main()
