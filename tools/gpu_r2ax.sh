#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2ax; mkdir -p $O
export TMPDIR=/tmp
for lib in lib/libzkmi355.so lib_exp/libzkmi355_cap32.so lib_exp/libzkmi355_cap24.so; do
echo "== $lib"
ZKMI355_LIB=$ROOT/zkevm-circuits_amd/$lib timeout 200 python tools/msm_runs.py 2000 2>&1 | grep "hint=2\|dense run=2000 hint=0" | cut -c1-140
ZKMI355_LIB=$ROOT/zkevm-circuits_amd/$lib timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/b.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/b.json')); print('   bench', d['value'], d['ms_per_step'], d['extra']['kernel_avg_ms']['msm_buckets'], d['extra']['kernel_avg_ms']['msm_combine'])"
done
