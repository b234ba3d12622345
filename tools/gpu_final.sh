#!/bin/bash
# what the driver runs at round end: GPU suite, smoke, the bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 900 python bench.py ) > $O/bench_full.json 2> $O/bench_full.err
python - <<PY
import json
d=json.load(open("$O/bench_full.json"))
print(d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", "frac", d["roofline"]["frac"], "lone", d["extra"]["msm_lone_ms"])
for k,v in d.get("proof",{}).items(): print(k, {x:v.get(x) for x in ("value","verified_by_oracle","error")})
print(d.get("cpu_baseline",{}).get("value"))
PY
tail -3 $O/bench_full.err
