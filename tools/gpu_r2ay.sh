#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2ay; mkdir -p $O
export TMPDIR=/tmp
ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --repeat 3 --no-verify > $O/sc.json 2> $O/sc.log
python -c "
import json; d=json.load(open('$O/sc.json')); print('pageable witness: sc shape', d['create_proof_s'], d['advice_host_memory'])"
grep "advice upload" $O/sc.log | tail -1
