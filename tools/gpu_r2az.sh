#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2az; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_proof.py tests/test_gpu_sharded_proof.py tests/test_gpu_msm.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|Error" $O/pytest.log | tail -3
ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 4 > $O/kc.json 2> $O/kc.log
echo "keccak $(python -c "import json; d=json.load(open('$O/kc.json')); print(d['create_proof_s'], d['verified_by_oracle'])") $(grep 'advice upload' $O/kc.log | tail -1)"
ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 > $O/sc.json 2> $O/sc.log
echo "sc $(python -c "import json; d=json.load(open('$O/sc.json')); print(d['create_proof_s'], d['verified_by_oracle'])") $(grep 'advice upload' $O/sc.log | tail -1)"
