#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2v; mkdir -p $O
export TMPDIR=/tmp
ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 22 --large --groups 3 --shplonk --pinned --repeat 3 --no-verify > $O/rec.json 2> $O/rec.log
python -c "
import json; d=json.load(open('$O/rec.json')); print('recursion shape', d['create_proof_s'], d['msm_count'])"
grep "zk prover" $O/rec.log | tail -34
ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 3 --no-verify > $O/kc.json 2> $O/kc.log
python -c "
import json; d=json.load(open('$O/kc.json')); print('keccak shape', d['create_proof_s'], d['msm_count'])"
grep "zk prover" $O/kc.log | grep -v "quotient:" | tail -17
