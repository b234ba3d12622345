#!/usr/bin/env python
"""Turns the rocprofv3 output of a record run (tools/gpu_r2_record.sh, earlier tools/gpu_r2z.sh; default gpurun_out/r2z/) into the committed summaries under
profiles/: kernel statistics of the bench line and of the SuperCircuit-shape proof, and the PMC
traffic (FETCH_SIZE / WRITE_SIZE, separate passes) per launch of the MSM and NTT kernels."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r2z"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
os.makedirs("profiles", exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("zk::", "")


def kernel_table(run, title, note):
    f = max(glob.glob(f"{src}/{run}/runc/*kernel_stats.csv"), key=os.path.getmtime)        # gpurun merges into gpurun_out/: an earlier pass may have left its files
    shutil.copy(f, f"profiles/{tag}_{run}_kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    out = [f"# {title}", "", note, "", "| kernel | calls | total ms | avg us | share |", "|---|---|---|---|---|"]
    for r in rows[:28]:
        out.append(f"| `{short(r['Name'])}` | {int(r['Calls'])} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} % |")
    return "\n".join(out) + "\n"


def pmc(run, counter):
    f = max(glob.glob(f"{src}/{run}/runc/*counter_collection.csv"), key=os.path.getmtime)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}          # KB per launch (rocprofv3 derived metric unit)


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
kern = ["k_msm_buckets", "k_msm_m_partition<20, false>", "k_msm_m_scatter_staged<20>", "k_msm_m_binsort", "k_ntt_pass", "k_ntt_last",
        "k_wsum_level<false>", "k_msm_combine_wave"]
traffic = {}
lines = ["# PMC traffic per launch (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, `bench.py --no-proof --no-cpu-baseline`)", "",
         "FETCH_SIZE / WRITE_SIZE are reported in KB at the L2 <-> fabric boundary.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE shows half of the bytes of a wide",
         "coalesced streaming read (16 B per lane) -- the `x2` column applies that correction, which is right for the streaming kernels (NTT passes,",
         "partition / bin passes) and an upper bound for the 64-byte gathers of `k_msm_buckets` (uncalibrated access width: both figures are given).", "",
         "| kernel | FETCH MiB | FETCH x2 MiB | WRITE MiB | FETCH + WRITE MiB | FETCH x2 + WRITE MiB |", "|---|---|---|---|---|---|"]
for k in kern:
    f_, w_ = fetch.get(k, 0.0) / 1024, write.get(k, 0.0) / 1024
    lines.append(f"| `{k}` | {f_:.1f} | {2 * f_:.1f} | {w_:.1f} | {f_ + w_:.1f} | {2 * f_ + w_:.1f} |")
    traffic[k] = {"fetch_MiB": round(f_, 1), "write_MiB": round(w_, 1)}
fb, wb = fetch.get("k_msm_buckets", 0.0) * 1024, write.get("k_msm_buckets", 0.0) * 1024
traffic_json = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py, see tools/gpu_r2z.sh",
                "msm_buckets_bytes_per_launch": int(fb + wb), "msm_buckets_fetch_bytes": int(fb), "msm_buckets_write_bytes": int(wb),
                "msm_buckets_bytes_per_launch_fetch_doubled": int(2 * fb + wb),
                "ntt_bytes_per_transform_fetch_doubled": int((2 * (fetch.get("k_ntt_pass", 0) + fetch.get("k_ntt_last", 0)) + write.get("k_ntt_pass", 0) + write.get("k_ntt_last", 0)) * 1024),
                "per_kernel": traffic}
json.dump(traffic_json, open(f"profiles/traffic_{tag}.json", "w"), indent=1)
lines += ["", f"`k_msm_buckets`: algorithmic bytes 96 B x 2^20 = 100.7 MB; counter traffic {(fb + wb) / 1e9:.2f} GB per launch (FETCH + WRITE as reported) = "
          f"{(fb + wb) / 100663296:.1f}x, round 1: 1.73 GB = 17.2x.  13 table gathers of 64 B per scalar are 872 MB of distinct data.",
          f"One 2^20 NTT: {traffic_json['ntt_bytes_per_transform_fetch_doubled'] / 2**20:.0f} MiB with the streaming correction against 64 MiB algorithmic "
          "(two passes plus the 32 MiB inter-pass twiddle table)."]
open(f"profiles/{tag}_pmc_traffic.md", "w").write("\n".join(lines) + "\n")
open(f"profiles/{tag}_bench_kernels.md", "w").write(kernel_table(
    "prof_bench", "Kernel statistics of the driver's bench command (`rocprofv3 --kernel-trace --stats -- python bench.py --no-proof --no-cpu-baseline`)",
    "48 timed + warm-up steps of MSM 2^20 + NTT 2^20 over 16 rotating columns, 6 lone commitments, SRS set-up (`k_fb_mul`, `k_build_window_tables` run once). "
    "`k_wsum_*` and `k_msm_reduce` run on the side streams under the next MSM."))
open(f"profiles/{tag}_scshape_kernels.md", "w").write(kernel_table(
    "prof_sc", "Kernel statistics of the SuperCircuit-shape proof (`bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2`)",
    "Two proofs + one keygen + the benchmark's own circuit construction (the 1 M-element `k_powers` launches build the identity permutation columns of the "
    "synthetic circuit on the device: data generation, not proving).  Kernel time sums over concurrent streams."))
print(open(f"profiles/{tag}_pmc_traffic.md").read())
