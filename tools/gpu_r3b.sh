#!/bin/bash
# round 3, pass b: sort-ahead A/B (bench line with and without) + the MSM / proof tests on the new loop
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_proof.py tests/test_gpu_params.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in 0 1; do
  ZK_MSM_SORT_AHEAD=$v timeout 600 python bench.py --steps 48 --warmup 16 --no-proof --no-cpu-baseline > $O/bench_sa$v.json 2> $O/bench_sa$v.err; echo "sort_ahead=$v rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/bench_sa$v.json").read())
print("value",d["value"],"ms/step",d["ms_per_step"],d["extra"])
PY
done
