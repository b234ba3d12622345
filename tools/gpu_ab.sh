#!/bin/bash
# A/B of library builds on the bench line: tools/gpu_ab.sh <label>=<path to .so> ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/ab; mkdir -p $O
export TMPDIR=/tmp
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  for i in 1 2; do
  ZKMI355_LIB=$ROOT/$lib timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench_$label.json 2> $O/bench_$label.err
  python -c "
import json; d=json.load(open('$O/bench_$label.json')); print('$label', d['value'], d['ms_per_step'], d['extra']['kernel_avg_ms'], 'lone', d['extra']['msm_lone_ms'])"
  done
done
