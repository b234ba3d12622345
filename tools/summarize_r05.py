#!/usr/bin/env python
"""Turns the output of tools/gpu_r5_record.sh (gpurun_out/r5rec/) into the committed summaries under profiles/: the bench line,
kernel statistics of the headline proof and of the MSM / NTT section (rocprofv3 --kernel-trace --stats), PMC traffic
(FETCH_SIZE / WRITE_SIZE, separate counter-only passes) per launch of the MSM / NTT kernels and per proof of the headline."""
import csv
import glob
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r5rec"
tag = "r05"
os.makedirs("profiles", exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("zk::", "")


def kernel_table(run, out_name, title, note, rows_max=32):
    f = max(glob.glob(f"{src}/{run}/runc/*kernel_stats.csv"), key=os.path.getmtime)
    shutil.copy(f, f"profiles/{tag}_{out_name}_kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    out = [f"# {title}", "", note, "", "| kernel | calls | total ms | avg us | share |", "|---|---|---|---|---|"]
    for r in rows[:rows_max]:
        out.append(f"| `{short(r['Name'])}` | {int(r['Calls'])} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} % |")
    open(f"profiles/{tag}_{out_name}_kernels.md", "w").write("\n".join(out) + "\n")
    agg = {}
    for r in rows:
        key = family(short(r["Name"]))
        c, t = int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6
        if key in agg:
            c, t = agg[key][0] + c, agg[key][1] + t
        agg[key] = (c, t, t * 1e3 / max(c, 1))
    return agg


def family(name):
    """the compile-time instances of the NTT passes (k_ntt_pass_f<10, true>, ...) count as their run-time namesakes"""
    for fam in ("k_ntt_pass", "k_ntt_last"):
        if name.startswith(fam):
            return fam
    return name


def sums(run):
    d = json.load(open(f"{src}/{run}_sums.json"))
    out = {}
    for k, v in d.items():
        key = family(k.split("|")[0])
        if key in out:
            out[key] = {"sum": out[key]["sum"] + v["sum"], "launches": out[key]["launches"] + v["launches"]}
        else:
            out[key] = dict(v)
    return out


bench = json.loads(open(f"{src}/bench_full.json").read().strip().splitlines()[-1])
json.dump(bench, open(f"profiles/bench_{tag}.json", "w"))
steps, warm = bench["steps"], bench["warmup"]
proofs_in_trace = steps + warm + 4          # + the two structure-blind proofs and the two host-memory proofs of the side measurements
kp = kernel_table("prof_proof", "proof", f"Kernel statistics of the headline run, round 5 (`bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify`)",
                  f"{proofs_in_trace} proofs of the SuperCircuit shape (k = 20, 60/30/10 witness, three phases: {warm} warm-up + {steps} timed with the witness resident in HBM, 2 with the structure-reading commitments off, 2 with the witness in "
                  "page-locked host memory) + one keygen + the benchmark's own circuit construction (`k_powers`, `k_scale`: data generation, not proving).  Kernel time sums over concurrent streams.")
km = kernel_table("prof_msmntt", "msmntt", "Kernel statistics of the MSM / NTT section, round 5 (`bench.py --only-msm-ntt --no-cpu-baseline`: BASELINE configs[1])",
                  "16 warm-up + 32 timed steps (one 2^20 commitment + one 2^20 transform each, batches of 32 columns) + 6 lone commitments; `k_build_window_tables` / `k_fb_mul` build the SRS and its tables once.")
fetch, write = sums("pmc_fetch"), sums("pmc_write")
fetch_p, write_p = sums("pmc_fetch_proof"), sums("pmc_write_proof")
KB = 1024.0
lines = [f"# PMC traffic, round 5 (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE: separate, counter-only passes; tools/gpu_r5_record.sh)", "",
         "FETCH_SIZE / WRITE_SIZE are reported in KB at the L2 <-> fabric boundary.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE shows half of the bytes of a wide coalesced streaming read",
         "(16 B per lane) -- the `x2` column applies that correction, which is right for the streaming kernels (NTT passes, partition / sort passes, the evaluator) and an upper",
         "bound for the 64-byte gathers of `k_msm_buckets` (uncalibrated access width: both figures are given).", "",
         "## MSM / NTT section (`bench.py --only-msm-ntt`): per launch", "",
         "| kernel | launches | FETCH MiB | FETCH x2 MiB | WRITE MiB | FETCH + WRITE MiB | FETCH x2 + WRITE MiB | algorithmic MiB |", "|---|---|---|---|---|---|---|---|"]
NTT_COLS = 16         # columns per launch of a 2^20 batch (csrc/ntt.hip: per_launch)
alg = {"k_msm_buckets": 96.0, "k_ntt_pass": NTT_COLS * 32.0, "k_ntt_last": NTT_COLS * 32.0}
per = {}
for k in ("k_msm_buckets", "k_msm_m_partition<20, false>", "k_msm_m_scatter_staged<20>", "k_msm_m_binsort", "k_ntt_pass", "k_ntt_last", "k_wsum_level<false>", "k_wsum_level<true>", "k_wsum_final"):
    if k not in fetch:
        continue
    n = fetch[k]["launches"]
    f_, w_ = fetch[k]["sum"] / n / KB, write.get(k, {"sum": 0, "launches": 1})["sum"] / max(write.get(k, {"launches": 1})["launches"], 1) / KB
    per[k] = (f_, w_, n)
    lines.append(f"| `{k}` | {n} | {f_:.1f} | {2 * f_:.1f} | {w_:.1f} | {f_ + w_:.1f} | {2 * f_ + w_:.1f} | {alg.get(k, '')} |")
lines += ["", "The NTT kernels carry SIXTEEN columns per launch (zk_ntt_batch; `k_ntt_pass` / `k_ntt_last` here = all compile-time instances `k_ntt_pass_f<..>` / `k_ntt_last_f<..>` together); a transform's algorithmic 64 MiB (read once, write once) are split over its two launches: 32 MiB per column and launch.",
          "`k_ntt_pass` reads the column and the 32 MiB inter-pass twiddle table (FETCH x2 = 2 x algorithmic), `k_ntt_last` reads the intermediate once.", "",
          f"## Headline proof (`bench.py --no-msm-ntt --steps 1 --warmup 0`: 5 proofs in the pass -- the timed one, two with the structure-reading commitments off, two from host memory --, figures per proof)", "",
          "| kernel | launches per proof | FETCH GiB | FETCH x2 GiB | WRITE GiB |", "|---|---|---|---|---|"]
proofs_pmc = 5.0
tot_f = tot_w = 0.0
for k, v in sorted(fetch_p.items(), key=lambda kv: -kv[1]["sum"])[:18]:
    f_ = v["sum"] / proofs_pmc / KB / KB
    w_ = write_p.get(k, {"sum": 0})["sum"] / proofs_pmc / KB / KB
    lines.append(f"| `{k}` | {v['launches'] / proofs_pmc:.0f} | {f_:.2f} | {2 * f_:.2f} | {w_:.2f} |")
for k, v in fetch_p.items():
    if k not in ("k_powers", "k_scale", "k_build_window_tables", "k_fb_mul", "k_fb_table"):
        tot_f += v["sum"]
        tot_w += write_p.get(k, {"sum": 0})["sum"]
lines += ["", f"All proving kernels of one proof: FETCH {tot_f / proofs_pmc / KB / KB:.1f} GiB (x2: {2 * tot_f / proofs_pmc / KB / KB:.1f}), WRITE {tot_w / proofs_pmc / KB / KB:.1f} GiB; "
          f"the line's `proof_roofline.algorithmic_bytes` = {bench['proof_roofline']['algorithmic_bytes'] / 2**30:.1f} GiB (halo2's full-extended-domain counts: the degree-class quotient reads fewer cosets)."]
open(f"profiles/{tag}_pmc_traffic.md", "w").write("\n".join(lines) + "\n")
mb = per["k_msm_buckets"]
nt = (per["k_ntt_pass"][0] * 2 + per["k_ntt_pass"][1] + per["k_ntt_last"][0] * 2 + per["k_ntt_last"][1]) / float(NTT_COLS)        # per transform, FETCH doubled (streaming reads)
q = fetch_p.get("k_quotient_eval<true>", {"sum": 0, "launches": 0})
traffic = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/gpu_r5_record.sh; profiles/r05_pmc_traffic.md",
    "msm_buckets_bytes_per_launch": int((mb[0] + mb[1]) * 2**20), "msm_buckets_fetch_bytes": int(mb[0] * 2**20), "msm_buckets_write_bytes": int(mb[1] * 2**20),
    "msm_buckets_bytes_per_launch_fetch_doubled": int((2 * mb[0] + mb[1]) * 2**20),
    "ntt_bytes_per_transform": int(nt * 2**20), "ntt_columns_per_launch": NTT_COLS,
    "ntt_bytes_per_transform_note": "FETCH x 2 (gfx950 streaming-read correction) + WRITE of k_ntt_pass and k_ntt_last, per column",
    "quotient_proof": {"launches": q["launches"] / proofs_pmc, "fetch_bytes_raw": int(q["sum"] / proofs_pmc * KB), "write_bytes": int(write_p.get("k_quotient_eval<true>", {"sum": 0})["sum"] / proofs_pmc * KB)},
    "proof_total": {"fetch_bytes_raw": int(tot_f / proofs_pmc * KB), "write_bytes": int(tot_w / proofs_pmc * KB)},
    "proof_traffic_bytes": {"fetch_raw": int(tot_f / proofs_pmc * KB), "fetch_x2": int(2 * tot_f / proofs_pmc * KB), "write": int(tot_w / proofs_pmc * KB),
                            "note": "all proving kernels of one headline proof (circuit construction excluded), mean over the five proofs of the counter passes; FETCH_SIZE raw and doubled (the guide's correction for wide streaming reads), WRITE_SIZE"},
}
json.dump(traffic, open(f"profiles/traffic_{tag}.json", "w"), indent=1)
print("bench value", bench["value"], "| msm buckets", round(mb[0] + mb[1], 1), "MiB per launch | ntt", round(nt, 1), "MiB per transform")
print("k_ntt_pass avg us", kp.get("k_ntt_pass", (0, 0, 0))[2], "k_ntt_last", kp.get("k_ntt_last", (0, 0, 0))[2], "| msm section k_msm_buckets avg us", km.get("k_msm_buckets", (0, 0, 0))[2])
