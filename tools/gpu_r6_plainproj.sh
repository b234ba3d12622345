#!/bin/bash
# the plain shape's worker alone (its line of `proof.*`, with the emulated rank of N = 2 / 4 / 8)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6plainproj; mkdir -p $O
timeout 900 python bench.py --proof-worker supercircuit_shape_k20_plain > $O/plain.json 2> $O/plain.err; echo rc=$?
python - $O/plain.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["value"], d["extra"].get("projected_rank_device_s"))
PY
tail -3 $O/plain.err
