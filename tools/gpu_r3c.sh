#!/bin/bash
# round 3, pass c: sort-ahead with a high-priority auxiliary stream
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
run() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py --steps 48 --warmup 16 --no-proof --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/bench_$tag.json").read())
print("value",d["value"],"ms/step",d["ms_per_step"],d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
PY
}
run sa0 ZK_MSM_SORT_AHEAD=0
run sa1_prio0 ZK_MSM_SORT_AHEAD=1 ZK_AUX_PRIORITY=0
run sa1_prio1 ZK_MSM_SORT_AHEAD=1 ZK_AUX_PRIORITY=1
