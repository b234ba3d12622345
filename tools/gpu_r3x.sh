#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
ZK_PROVER_TRACE=1 timeout 600 python bench.py --proof-worker supercircuit_shape_k20 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('sc', d['value'], d.get('create_proof_s'))"
grep "zk prover" /tmp/err.txt | tail -75 | grep -v "quotient: " | awk '/shplonk: /{a[$4" "$5]+=$(NF-1); next} {print} END {for (k in a) print "   sum shplonk:", k, a[k]}'
