#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2p; mkdir -p $O
export TMPDIR=/tmp
ZK_PROVER_TRACE=1 timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep "zk prover" $O/bench.err | grep -v "quotient:" | tail -18
ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 --no-verify > $O/sc.json 2> $O/sc.log
echo standalone; grep "zk prover" $O/sc.log | grep -v "quotient:" | tail -18
python -c "
import json; d=json.load(open('$O/sc.json')); print('standalone', d['create_proof_s'])
d=json.load(open('$O/bench.json')); print('in bench', d['proof']['supercircuit_shape_k20']['create_proof_s'])"
