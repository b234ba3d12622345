#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2t; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python bench.py ) > $O/bench_full.json 2> $O/bench_full.err
python - <<PY
import json
d=json.load(open("$O/bench_full.json"))
print(d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
for k,v in d.get("proof",{}).items(): print(k, {x:v.get(x) for x in ("value","create_proof_s","verified_by_oracle","error","chain_of_4_proofs_s")}, (v.get("roofline_quotient") or {}).get("frac"))
print(d.get("cpu_baseline"))
PY
tail -4 $O/bench_full.err
