#!/bin/bash
# round 3, pass o: the driver's bench command (MSM + NTT line, CPU baseline, three proof shapes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3o; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value",d["value"],"ms/step",d["ms_per_step"],"lone",d["extra"]["msm_lone_ms"])
for k,v in d.get("proof",{}).items():
    print(k, v.get("value"), v.get("create_proof_s"), "verified", v.get("verified_by_oracle"), "quot frac", (v.get("roofline_quotient") or {}).get("frac"))
PY
