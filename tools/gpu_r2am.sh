#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2am; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python bench.py ) > $O/bench_full.json 2> $O/bench_full.err
python - <<PY
import json
d=json.load(open("$O/bench_full.json"))
print(d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"], "traffic", d["roofline"]["traffic"])
for k,v in d.get("proof",{}).items(): print(k, {x:v.get(x) for x in ("value","create_proof_s","verified_by_oracle","error","chain_of_4_proofs_s")}, (v.get("roofline_quotient") or {}).get("frac"))
PY
tail -3 $O/bench_full.err
timeout 600 python bench.py --gpus 2 --steps 8 --warmup 4 > $O/bench2.json 2> $O/bench2.err; echo "bench --gpus 2 rc=$?"
python -c "
import json; d=json.load(open('$O/bench2.json')); print(d['n_gpus'], d['value'], d.get('proof_sharded'))"
