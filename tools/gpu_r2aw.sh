#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2aw; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_field.py tests/test_gpu_proof.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -1
ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 --no-verify > $O/sc.json 2> $O/sc.log
python -c "
import json; d=json.load(open('$O/sc.json')); print('sc shape', d['create_proof_s'])"
grep "lookup: m\|lookup: phi\|lookup m \|lookup phi " $O/sc.log | tail -4
