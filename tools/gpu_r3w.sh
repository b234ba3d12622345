#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3w; mkdir -p $O
export TMPDIR=/tmp
ZK_PROVER_TRACE=1 timeout 600 python bench_proof.py --large --k 22 --groups 3 --shplonk --pinned --repeat 3 --no-verify > $O/rec.json 2> $O/rec.err
tail -1 $O/rec.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('recursion k22', d['create_proof_s'])"
grep "zk prover" $O/rec.err | tail -34
ZK_PROVER_TRACE=1 timeout 600 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 3 --no-verify > $O/kc.json 2> $O/kc.err
tail -1 $O/kc.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('keccak k18', d['create_proof_s'])"
grep "zk prover" $O/kc.err | grep -v "quotient: " | tail -22; grep "quotient: " $O/kc.err | tail -16 | awk '{a[$4]+=$(NF-1)} END {for (k in a) print "  quotient", k, a[k]}'
