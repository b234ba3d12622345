#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2af; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_proof.py tests/test_gpu_comm.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|Error" $O/pytest.log | tail -3
for i in 1 2; do
  timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench.json 2> $O/bench.err
  python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['extra']['kernel_avg_ms'], 'lone', d['extra']['msm_lone_ms'])"
done
timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 --no-verify > $O/sc.json 2> $O/sc.log
python -c "
import json; d=json.load(open('$O/sc.json')); print('sc shape', d['create_proof_s'])"
