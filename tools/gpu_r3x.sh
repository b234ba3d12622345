#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
true
timeout 300 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 4 --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('keccak k18', d['create_proof_s'])"
timeout 600 python bench.py --proof-worker supercircuit_shape_k20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('sc', d['value'], d.get('create_proof_s'))"
