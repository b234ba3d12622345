#!/usr/bin/env python3
"""Device idle time out of a rocprofv3 kernel trace (csv): the union of all kernel intervals against the span they cover, idle gaps
attributed to the kernel that ran before them, and -- between two marker kernels -- the same for one window (a headline proof runs
between two `k_sample_large` launches of consecutive advice phases of the same kind).

    python tools/kernel_gaps.py <kernel_trace.csv> [min_gap_us]
"""
import collections, csv, sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zk::", "")))
rows.sort()
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 5e3
# busy union
busy, cur_s, cur_e = 0, None, None
gaps = []          # (gap ns, kernel before, kernel after, time)
last_name = None
for s, e, name in rows:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        if e > cur_e:
            cur_e = e
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, name, cur_e))
        cur_s, cur_e = s, e
    if last_name is None or e >= cur_e:
        last_name = name
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)}, span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %)")
# drop the long set-up gaps (host-side circuit construction, verification): gaps above 20 ms
work_gaps = [g for g in gaps if g[0] < 20e6]
idle = sum(g[0] for g in work_gaps)
print(f"idle in gaps below 20 ms: {idle / 1e6:.1f} ms in {len(work_gaps)} gaps; above: {sum(g[0] for g in gaps if g[0] >= 20e6) / 1e6:.1f} ms in {sum(1 for g in gaps if g[0] >= 20e6)}")
by = collections.defaultdict(lambda: [0, 0])
for g, before, after, t in work_gaps:
    by[(before, after)][0] += g
    by[(before, after)][1] += 1
print("idle by (kernel before -> kernel after), top 25:")
for (b, a), (tot, cnt) in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {tot / 1e6:8.2f} ms  {cnt:6d} gaps  avg {tot / cnt / 1e3:7.1f} us   {b[:38]:38s} -> {a[:38]}")
hist = collections.Counter()
for g, *_ in work_gaps:
    hist[min(int(g / 1e3) // 10 * 10, 200)] += g
print("idle by gap length (us bucket: ms):", ", ".join(f"{k}+: {v / 1e6:.1f}" for k, v in sorted(hist.items())))
