#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2m; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_quotient.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench_$label.json 2> $O/bench_$label.err
  python - <<PY
import json
d=json.load(open("$O/bench_$label.json"))
print("$label", d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
PY
}
run bs1_ts5 ZK_MSM_BINSORT=1
run bs0_ts5 ZK_MSM_BINSORT=0
run bs1_ts0 ZK_MSM_BINSORT=1 ZK_MSM_TOP_SHIFT=0
run bs0_ts0 ZK_MSM_BINSORT=0 ZK_MSM_TOP_SHIFT=0
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $ROOT/bench.py --no-proof --no-cpu-baseline > $O/prof.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:22]:
        print(f'{r["Name"][:60]:60s} {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
