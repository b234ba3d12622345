"""Reads the A/B measurements of tools/gpu_arith_ab.sh and names the build to keep.
  pick  DIR          -> cMNQ: per group (M: msm + ecntt, N: ntt, Q: quotient + vec) 1 if the carry-first build (c111) beat the
                        compiler-associated one (c000) on that group's own timing by more than the run-to-run noise
  final DIR CHOSEN   -> CHOSEN if its bench value beats head's by more than the noise, else head
"""
import json
import re
import sys

NOISE = 0.003


def bench(d, v):
    with open(f"{d}/bench_{v}.json") as f:
        return json.loads(f.read().strip().splitlines()[-1])


def quot_ms(d, v):
    m = re.search(r":\s*([0-9.]+) ms per launch", open(f"{d}/quot_{v}.txt").read())
    return float(m.group(1))


def main():
    mode, d = sys.argv[1], sys.argv[2]
    if mode == "pick":
        a, b = bench(d, "c000"), bench(d, "c111")
        m = int(b["extra"]["msm_pipelined_ms"] < a["extra"]["msm_pipelined_ms"] * (1 - NOISE))
        n = int(b["extra"]["ntt_only_ms"] < a["extra"]["ntt_only_ms"] * (1 - NOISE))
        try:
            q = int(quot_ms(d, "c111") < quot_ms(d, "c000") * (1 - NOISE))
        except Exception:
            q = 0
        print(f"c{m}{n}{q}")
    elif mode == "final":
        ch = sys.argv[3]
        try:
            better = bench(d, ch)["value"] > bench(d, "head")["value"] * (1 + NOISE)
        except Exception:
            better = False
        print(ch if better else "head")
    else:
        raise SystemExit("pick | final")


if __name__ == "__main__":
    main()
