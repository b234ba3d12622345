#!/usr/bin/env python
"""Stage table of ONE proof out of a `ZK_PROVER_TRACE=1` log (stderr of a prover process): the top-level marks in order with
their share of the proof, the indented sub-marks summed per label (the quotient's per-class lines, the SHPLONK steps per
rotation set).  The finds of round 3 -- the SHPLONK basis rebuilt per column, the host-side sampling in front of the uploads,
lookup tables hashed once per argument -- all came out of tables like this one.

usage: ZK_PROVER_TRACE=1 python bench_proof.py ... 2> trace.log ; python tools/trace_stages.py trace.log [proof index from the end, default 1]"""
import collections
import re
import sys

LINE = re.compile(r"^\[zk prover\] (\s*)(.*?)\s+([0-9]+\.[0-9]+) ms\s*$")


def proofs(lines):
    """split the marks into proofs: a proof ends with its multi-open mark"""
    cur, out = [], []
    for ln in lines:
        m = LINE.match(ln.rstrip("\n"))
        if not m:
            continue
        indent, label, ms = len(m.group(1)), m.group(2), float(m.group(3))
        cur.append((indent > 0, label, ms))
        if not indent and label.startswith("multiopen"):
            out.append(cur)
            cur = []
    return out


def table(marks):
    total = sum(ms for sub, _, ms in marks)
    rows = []
    pending = collections.OrderedDict()          # sub-marks since the last top-level mark, summed per label
    for sub, label, ms in marks:
        if sub:
            key = re.sub(r"\s+", " ", label)
            pending[key] = pending.get(key, 0.0) + ms
            continue
        inner = sum(pending.values())
        rows.append((label, ms + inner, list(pending.items()), ms))
        pending = collections.OrderedDict()
    out = [f"{'stage':44s} {'ms':>9s} {'share':>7s}"]
    for label, ms, subs, own in rows:
        out.append(f"{label:44s} {ms:9.2f} {100 * ms / total:6.1f} %")
        for k, v in subs:
            out.append(f"    {k:40s} {v:9.2f}")
        if subs and own > 0.005:
            out.append(f"    {'(rest of the stage)':40s} {own:9.2f}")
    out.append(f"{'sum of the marks':44s} {total:9.2f}")
    return "\n".join(out)


def main():
    ps = proofs(open(sys.argv[1], errors="ignore").read().splitlines())
    if not ps:
        raise SystemExit("no complete proof in this log (was ZK_PROVER_TRACE=1 set?)")
    idx = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print(f"proof {len(ps) - idx + 1} of {len(ps)} in the log")
    print(table(ps[-idx]))


if __name__ == "__main__":
    main()
