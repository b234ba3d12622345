#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2at; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_sharded_ntt.py tests/test_gpu_proof.py tests/test_gpu_params.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -1
timeout 300 python tools/ntt_sizes.py 2>&1 | cut -c1-100
timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 --no-verify > $O/sc.json 2> $O/sc.log
python -c "
import json; d=json.load(open('$O/sc.json')); print('sc shape', d['create_proof_s'])"
timeout 400 python bench_proof.py --k 22 --large --groups 3 --shplonk --pinned --repeat 3 --no-verify > $O/rec.json 2> $O/rec.log
python -c "
import json; d=json.load(open('$O/rec.json')); print('recursion shape', d['create_proof_s'])"
