#!/usr/bin/env python
"""One steady-state MSM period out of a rocprofv3 --kernel-trace CSV: every kernel between two consecutive
k_msm_buckets launches with its stream, start offset, duration and the gap to the previous kernel on the same
stream.  Usage: tools/trace_timeline.py <kernel_trace.csv> [index of the k_msm_buckets launch, default 40]"""
import collections
import csv
import sys


def short(n):
    return n.split("(")[0].replace("void ", "").replace("zk::", "")[:36]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    idx = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    t0 = int(rows[0]["Start_Timestamp"])
    by = collections.defaultdict(list)
    for r in rows:
        by[short(r["Kernel_Name"])].append(r)
    for n in ("k_msm_combine_wave", "k_msm_combine_small", "k_msm_combine", "k_size_bins_scan", "k_msm_buckets"):
        if n not in by:
            continue
        d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in by[n])
        r = by[n][0]
        print(f"{n}: grid {r['Grid_Size_X']} vgpr {r['VGPR_Count']} scratch {r['Scratch_Size']}  min {d[0]:.1f} med {d[len(d) // 2]:.1f} max {d[-1]:.1f} us")
    bk = by["k_msm_buckets"]
    s40, s41 = int(bk[idx]["Start_Timestamp"]), int(bk[idx + 1]["Start_Timestamp"])
    print(f"period {(s41 - s40) / 1e3:.1f} us, main stream {bk[idx]['Stream_Id']}")
    seq = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], short(r["Kernel_Name"])) for r in rows
                 if s40 <= int(r["Start_Timestamp"]) < s41)
    prev_end = {}
    for s, e, st, n in seq:
        gap = (s - prev_end.get(st, s)) / 1e3
        print(f"{(s - s40) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  st{st}  gap {gap:6.1f}  {n}")
        prev_end[st] = e


if __name__ == "__main__":
    main()
