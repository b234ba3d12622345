#!/bin/bash
# A/B of the headline proof on ONE box, variants alternating (box-to-box and run-to-run spread is ~1 %: single runs do not resolve less)
# usage: tools/gpu_ab.sh <tag> <steps> <rounds> "<env of A>" "<env of B>" ["<env of C>" ...]      (an env is a space-separated list of VAR=value, "-" for none)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); TAG=$1; STEPS=$2; ROUNDS=$3; shift 3
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for r in $(seq 1 $ROUNDS); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    envs=""; [ "$v" != "-" ] && envs="$v"
    env $envs ZK_BENCH_QUICK=1 timeout 600 python bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify --steps $STEPS --warmup 1 > $O/v${i}_r$r.json 2> $O/v${i}_r$r.err
    python - "$O/v${i}_r$r.json" "variant $i [$v] round $r" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e = d["extra"]
    print(sys.argv[2], "value", d["value"], "classes", e["kernel_class_device_ms_per_proof"])
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
  done
done
python - $O "$@" <<'PY'
import glob, json, sys
O = sys.argv[1]
for i, v in enumerate(sys.argv[2:], 1):
    vals = []
    for f in sorted(glob.glob(f"{O}/v{i}_r*.json")):
        try: vals.append(json.loads([l for l in open(f) if l.startswith("{")][-1])["value"])
        except Exception: pass
    if vals: print(f"variant {i} [{v}]: mean {sum(vals) / len(vals):.4f} s, min {min(vals):.4f}, runs {vals}")
PY
