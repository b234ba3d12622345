#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2n; mkdir -p $O
export TMPDIR=/tmp
run() { local label=$1; shift
  env "$@" timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench_$label.json 2> $O/bench_$label.err
  python - <<PY
import json
d=json.load(open("$O/bench_$label.json"))
print("$label", d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
PY
}
run chunk1024 ZK_MSM_CHUNK=1024
run chunk2048 ZK_MSM_CHUNK=2048
run chunk4096 ZK_MSM_CHUNK=4096
run chunk8192 ZK_MSM_CHUNK=8192
