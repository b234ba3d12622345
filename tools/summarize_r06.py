#!/usr/bin/env python
"""Turns the output of tools/gpu_r6_record.sh (gpurun_out/r6rec/) into the committed summaries under profiles/: the bench line,
kernel statistics PER PROOF KIND of the EVM-style headline (timed / structure-blind / degree-blind / host-memory) and of the plain
shape (rocprofv3 --kernel-trace --stats), of the MSM / NTT section, PMC traffic (FETCH_SIZE / WRITE_SIZE, separate counter-only
passes) per launch of the MSM / NTT kernels and per proof of the headline, and the issue counters of one whole proof."""
import csv
import glob
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r6rec"
tag = "r06"
os.makedirs("profiles", exist_ok=True)
SETUP = ("k_powers", "k_scale", "k_build_window_tables", "k_fb_mul", "k_fb_table")          # the benchmark's circuit construction and the SRS tables: not proving


def short(name):
    return name.split("(")[0].replace("void ", "").replace("zk::", "")


def family(name):
    """the compile-time instances of the NTT passes (k_ntt_pass_f<10, true>, ...) count as their run-time namesakes"""
    for fam in ("k_ntt_pass", "k_ntt_last", "k_quotient_eval"):
        if name.startswith(fam):
            return fam
    return name


def kernel_table(run, out_name, title, note, rows_max=28):
    files = glob.glob(f"{src}/{run}/**/*kernel_stats.csv", recursive=True)
    if not files:
        return {}
    f = max(files, key=os.path.getmtime)
    shutil.copy(f, f"profiles/{tag}_{out_name}_kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    out = [f"# {title}", "", note, "", "| kernel | calls | total ms | avg us | share |", "|---|---|---|---|---|"]
    for r in rows[:rows_max]:
        out.append(f"| `{short(r['Name'])}` | {int(r['Calls'])} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} % |")
    agg = {}
    for r in rows:
        key = family(short(r["Name"]))
        c, t = int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6
        if key in agg:
            c, t = agg[key][0] + c, agg[key][1] + t
        agg[key] = (c, t, t * 1e3 / max(c, 1))
    classes = {"NTT passes (k_ntt_pass*, k_ntt_last*)": ("k_ntt_pass", "k_ntt_last"), "evaluator (k_quotient_eval*)": ("k_quotient_eval",),
               "MSM bucket accumulation (k_msm_buckets)": ("k_msm_buckets",)}
    out += ["", "| kernel class | calls | total ms |", "|---|---|---|"]
    for label, fams in classes.items():
        c = sum(agg.get(f_, (0, 0, 0))[0] for f_ in fams)
        t = sum(agg.get(f_, (0, 0, 0))[1] for f_ in fams)
        out.append(f"| {label} | {c} | {t:.1f} |")
    msm_side = sum(v[1] for k_, v in agg.items() if k_.startswith(("k_wsum", "k_msm_reduce", "k_msm_window_sum")))
    msm_sort = sum(v[1] for k_, v in agg.items() if k_.startswith(("k_msm_gm_partition", "k_msm_m_partition", "k_msm_m_scatter", "k_msm_m_binsort")))
    prov = sum(v[1] for k_, v in agg.items() if k_ not in SETUP)
    out += [f"| MSM sorts (partition / scatter / binsort) | | {msm_sort:.1f} |", f"| MSM side kernels (weighted sums, reductions) | | {msm_side:.1f} |", f"| all proving kernels (circuit construction excluded) | | {prov:.1f} |"]
    open(f"profiles/{tag}_{out_name}_kernels.md", "w").write("\n".join(out) + "\n")
    return agg


def sums(run):
    p = f"{src}/{run}_sums.json"
    if not os.path.exists(p):
        return {}
    d = json.load(open(p))
    out = {}
    for k, v in d.items():
        name, ctr = k.split("|")
        key = (family(name), ctr)
        if key in out:
            out[key] = {"sum": out[key]["sum"] + v["sum"], "launches": out[key]["launches"] + v["launches"]}
        else:
            out[key] = dict(v)
    return out


bench = json.loads(open(f"{src}/bench_full.json").read().strip().splitlines()[-1])
json.dump(bench, open(f"profiles/bench_{tag}.json", "w"))
kinds = {"timed": "the timed kind: witness resident in HBM, every path on", "structure_blind": "`ZK_MSM_RUNS=0 ZK_MSM_DIFF=0`: permutation products and lookup sums committed as dense columns",
         "degree_blind": "`ZK_QUOTIENT_SPLIT=0 ZK_QUOTIENT_ADDSPLIT=0`: one degree class, every column on all 8 cosets", "host": "`ZK_BENCH_KIND=host`: the witness in page-locked host memory (zk_proof_advice_phase)",
         "plain": "`ZK_BENCH_SHAPE=plain`: rounds 1-5's shape (one degree-9 gate on three columns), timed kind"}
per_kind = {}
for kind, what in kinds.items():
    per_kind[kind] = kernel_table(f"prof_{kind}", f"proof_{kind}", f"Kernel statistics of ONE proof kind, round 6: {kind} (`bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify --steps 1 --warmup 1`, side measurements off)",
                                  f"2 proofs (1 warm-up + 1 timed) of the {'plain' if kind == 'plain' else 'EVM-style'} SuperCircuit shape at k = 20, three phases -- {what} -- + one keygen + the benchmark's own circuit "
                                  "construction (`k_powers`, `k_scale`: data generation, not proving).  Kernel time sums over concurrent streams; divide by 2 for one proof.")
km = kernel_table("prof_msmntt", "msmntt", "Kernel statistics of the MSM / NTT section, round 6 (`bench.py --only-msm-ntt --no-cpu-baseline`: BASELINE configs[1])",
                  "16 warm-up + 32 timed steps (one 2^20 commitment + one 2^20 transform each, batches of 32 columns) + 6 lone commitments; `k_build_window_tables` / `k_fb_mul` build the SRS and its tables once.")
fetch, write = sums("pmc_fetch"), sums("pmc_write")
fetch_p, write_p = sums("pmc_fetch_proof"), sums("pmc_write_proof")
KB = 1024.0
lines = ["# PMC traffic, round 6 (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE: separate, counter-only passes; tools/gpu_r6_record.sh)", "",
         "FETCH_SIZE / WRITE_SIZE are reported in KB at the L2 <-> fabric boundary.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE shows half of the bytes of a wide coalesced streaming read",
         "(16 B per lane) -- the `x2` column applies that correction, which is right for the streaming kernels (NTT passes, partition / sort passes, the evaluator's operand loads) and an upper",
         "bound for the 64-byte gathers of `k_msm_buckets` (uncalibrated access width: both figures are given).", "",
         "## MSM / NTT section (`bench.py --only-msm-ntt`): per launch", "",
         "| kernel | launches | FETCH MiB | FETCH x2 MiB | WRITE MiB | FETCH + WRITE MiB | FETCH x2 + WRITE MiB | algorithmic MiB |", "|---|---|---|---|---|---|---|---|"]
NTT_COLS = 16
alg = {"k_msm_buckets": 96.0, "k_ntt_pass": NTT_COLS * 32.0, "k_ntt_last": NTT_COLS * 32.0}
per = {}
for k in ("k_msm_buckets", "k_msm_m_partition<20, false>", "k_msm_m_scatter_staged<20>", "k_msm_m_binsort", "k_ntt_pass", "k_ntt_last", "k_wsum_level<false>", "k_wsum_level<true>", "k_wsum_final"):
    if (k, "FETCH_SIZE") not in fetch:
        continue
    n = fetch[(k, "FETCH_SIZE")]["launches"]
    wv = write.get((k, "WRITE_SIZE"), {"sum": 0, "launches": 1})
    f_, w_ = fetch[(k, "FETCH_SIZE")]["sum"] / n / KB, wv["sum"] / max(wv["launches"], 1) / KB
    per[k] = (f_, w_, n)
    lines.append(f"| `{k}` | {n} | {f_:.1f} | {2 * f_:.1f} | {w_:.1f} | {f_ + w_:.1f} | {2 * f_ + w_:.1f} | {alg.get(k, '')} |")
lines += ["", "The NTT kernels carry SIXTEEN columns per launch; a transform's algorithmic 64 MiB (read once, write once) are split over its two launches: 32 MiB per column and launch.", "",
          "## EVM-style headline proof (`bench.py --no-msm-ntt --steps 1 --warmup 0`, side measurements off: ONE proof in the pass)", "",
          "| kernel | launches | FETCH GiB | FETCH x2 GiB | WRITE GiB |", "|---|---|---|---|---|"]
tot_f = tot_w = 0.0
for (k, c), v in sorted(fetch_p.items(), key=lambda kv: -kv[1]["sum"])[:18]:
    f_ = v["sum"] / KB / KB
    w_ = write_p.get((k, "WRITE_SIZE"), {"sum": 0})["sum"] / KB / KB
    lines.append(f"| `{k}` | {v['launches']} | {f_:.2f} | {2 * f_:.2f} | {w_:.2f} |")
for (k, c), v in fetch_p.items():
    if k not in SETUP:
        tot_f += v["sum"]
        tot_w += write_p.get((k, "WRITE_SIZE"), {"sum": 0})["sum"]
qf = fetch_p.get(("k_quotient_eval", "FETCH_SIZE"), {"sum": 0, "launches": 0})
qw = write_p.get(("k_quotient_eval", "WRITE_SIZE"), {"sum": 0, "launches": 0})
ev = bench["extra"]["evaluator"]
q_exec = None
fetched_share = lambda loaded: 100.0 * 2 * qf["sum"] * KB / loaded            # FETCH_SIZE is in KB; doubled as the guide prescribes
for r in bench["rooflines"]:
    if "k_quotient_eval" in r["kernel"]:
        q_exec = r.get("executed")
lines += ["", f"All proving kernels of the proof: FETCH {tot_f / KB / KB:.1f} GiB (x2: {2 * tot_f / KB / KB:.1f}), WRITE {tot_w / KB / KB:.1f} GiB; "
          f"the line's `proof_roofline.algorithmic_bytes` = {bench['proof_roofline']['algorithmic_bytes'] / 2**30:.1f} GiB (halo2's full-extended-domain counts).",
          f"The evaluator (all launches of the proof, compressions and linear combinations included): FETCH {qf['sum'] / KB / KB:.1f} GiB raw = {2 * qf['sum'] / KB / KB:.1f} GiB corrected, WRITE {qw['sum'] / KB / KB:.1f} GiB"
          + (f"; the class programs alone execute {q_exec['operand_bytes_per_proof'] / 2**30:.1f} GiB of operand loads per proof (`rooflines[..].executed`): {fetched_share(q_exec['operand_bytes_per_proof']):.0f} % of what the interpreter loads is fetched from beyond the L2 -- the first record run of the round read 88 % (every load a miss); the slices of a sliced class program (csrc/quotient.hip: plan_slices) find each other's operands in the caches." if q_exec else ".")]
open(f"profiles/{tag}_pmc_traffic.md", "w").write("\n".join(lines) + "\n")
traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/gpu_r6_record.sh; profiles/r06_pmc_traffic.md"}
if "k_msm_buckets" in per and "k_ntt_pass" in per:
    mb = per["k_msm_buckets"]
    nt = (per["k_ntt_pass"][0] * 2 + per["k_ntt_pass"][1] + per["k_ntt_last"][0] * 2 + per["k_ntt_last"][1]) / float(NTT_COLS)
    traffic.update({"msm_buckets_bytes_per_launch": int((mb[0] + mb[1]) * 2**20), "msm_buckets_fetch_bytes": int(mb[0] * 2**20), "msm_buckets_write_bytes": int(mb[1] * 2**20),
                    "msm_buckets_bytes_per_launch_fetch_doubled": int((2 * mb[0] + mb[1]) * 2**20), "ntt_bytes_per_transform": int(nt * 2**20), "ntt_columns_per_launch": NTT_COLS,
                    "ntt_bytes_per_transform_note": "FETCH x 2 (gfx950 streaming-read correction) + WRITE of k_ntt_pass and k_ntt_last, per column"})
if qf["launches"]:
    coset_launches = ev["class_launches_per_proof"]
    traffic["quotient_bytes_per_launch"] = int((2 * qf["sum"] + qw["sum"]) * KB / max(coset_launches, 1))
    traffic["quotient_bytes_per_proof"] = int((2 * qf["sum"] + qw["sum"]) * KB)
    traffic["quotient_note"] = (f"FETCH x 2 (the guide's correction for 16-byte-per-lane streaming reads) + WRITE of every k_quotient_eval launch of ONE EVM-style headline proof ({qf['launches']} launches), "
                                f"divided by the {coset_launches:.0f} degree-class launches that `roofline.avg_launch_ms` averages over (the other launches are the proof's compressions and linear "
                                "combinations: a percent of the bytes)")
traffic["proof_traffic_bytes"] = {"fetch_raw": int(tot_f * KB), "fetch_x2": int(2 * tot_f * KB), "write": int(tot_w * KB),
                                  "note": "all proving kernels of ONE EVM-style headline proof (circuit construction excluded); FETCH_SIZE raw and doubled (the guide's correction for wide streaming reads), WRITE_SIZE"}
json.dump(traffic, open(f"profiles/traffic_{tag}.json", "w"), indent=1)

# ---- issue counters of one whole proof
issue, waves = sums("pmc_issue_proof"), sums("pmc_waves_proof")
ctrs = ("SQ_INSTS_VALU", "SQ_INSTS_VALU_INT64", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_SALU", "SQ_INSTS_LDS")
by_kernel = {}
for d in (issue, waves):
    for (k, c), v in d.items():
        if k in SETUP:
            continue
        by_kernel.setdefault(k, {})[c] = v["sum"]
tot = {c: sum(v.get(c, 0) for v in by_kernel.values()) for c in ctrs}
out = ["# Issue counters of ONE whole EVM-style headline proof, round 6 (`rocprofv3 --pmc`, two counter-only passes over `bench.py --no-msm-ntt --steps 1 --warmup 0`, side measurements off)", "",
       "Sums over every proving kernel of the proof (circuit construction excluded).  `SQ_INSTS_VALU_INT64` counts the 64-bit integer multiply-adds: the useful products are 162 of them each.", "",
       "| counter | whole proof |", "|---|---|"]
for c in ctrs:
    out.append(f"| {c} | {tot[c]:.4g} |")
if tot["SQ_INSTS_VALU"]:
    out += ["", f"Multiply-adds are {100 * tot['SQ_INSTS_VALU_INT64'] / tot['SQ_INSTS_VALU']:.1f} % of the vector instructions of a proof = {tot['SQ_INSTS_VALU_INT64'] / 162 / 1e9:.1f} G wave-level product-equivalents x 64 lanes = "
            f"{tot['SQ_INSTS_VALU_INT64'] * 64 / 162 / 1e9:.0f} G field products; at {bench['value']:.3f} s per proof that is {tot['SQ_INSTS_VALU_INT64'] * 64 / 162 / 1e9 / bench['value']:.0f} G products/s "
            "against the 169 G/s of the product routine alone."]
out += ["", "| kernel | SQ_INSTS_VALU | of them multiply-adds | share of the proof's vector instructions |", "|---|---|---|---|"]
for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:16]:
    va = v.get("SQ_INSTS_VALU", 0)
    if va:
        out.append(f"| `{k}` | {va:.4g} | {100 * v.get('SQ_INSTS_VALU_INT64', 0) / va:.1f} % | {100 * va / max(tot['SQ_INSTS_VALU'], 1):.1f} % |")
open(f"profiles/{tag}_proof_pmc.md", "w").write("\n".join(out) + "\n")
print("bench value", bench["value"], "| kinds", {k_: round(sum(v[1] for n_, v in a.items() if n_ not in SETUP), 1) for k_, a in per_kind.items()})
