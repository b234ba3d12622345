#!/bin/bash
# round 3, pass e: few-task buckets combined inside k_msm_buckets -- MSM / proof tests, bench line, narrow-column timing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_proof.py tests/test_gpu_params.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 600 python bench.py --steps 48 --warmup 16 --no-proof --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read())
print("value",d["value"],"ms/step",d["ms_per_step"],d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
PY
ZK_BENCH_SCALAR_BITS=30 timeout 600 python bench.py --steps 48 --warmup 16 --no-proof --no-cpu-baseline > $O/bench30.json 2> $O/bench30.err; echo "bench30 rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench30.json").read())
print("30-bit value",d["value"],"ms/step",d["ms_per_step"],d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
PY
