#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for v in 1 0 1 0; do
  ZK_LOOKUP_SHARE=$v ZK_PROVER_TRACE=1 timeout 600 python bench.py --proof-worker supercircuit_shape_k20 2>/tmp/err_$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('share=$v sc', d['value'], d.get('create_proof_s'))"
  grep "lookup" /tmp/err_$v.txt | grep "zk prover" | tail -4
done
