#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_msm.py -q --tb=short -p no:cacheprovider ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 200 python bench.py --no-proof --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json")); e=d["extra"]
print(d["value"], "Mscalar/s", d["ms_per_step"], "ms/step lone", e["msm_lone_ms"], e["kernel_avg_ms"])
PY
