#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2aa; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_proof.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|Error" $O/pytest.log | tail -3
for st in 1 0 1 0; do
  ZK_MSM_STAGED=$st timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench_$st.json 2> $O/bench_$st.err
  python -c "
import json; d=json.load(open('$O/bench_$st.json')); print('staged=$st', d['value'], d['ms_per_step'], d['extra']['kernel_avg_ms'], 'lone', d['extra']['msm_lone_ms'])"
done
