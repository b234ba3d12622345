#!/bin/bash
# round 3, pass g: is the device-scope fence what the in-kernel combination costs?  (nofence build = measurement only, not correct)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3g; mkdir -p $O
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
run fence ZK_X=1
run nofence ZKMI355_LIB=$ROOT/zkevm-circuits_amd/lib/libzkmi355_nofence.so
run fence2 ZK_X=1
