#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_proof.py tests/test_gpu_sharded_proof.py -x -q -m gpu 2>&1 | tail -2
ZK_PROVER_TRACE=1 timeout 300 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 3 --no-verify 2>/tmp/kc.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('keccak k18 (traced)', d['create_proof_s'])"
grep "zk prover" /tmp/kc.err | tail -30 | grep -A6 "h recombination" 
timeout 300 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('keccak k18', d['create_proof_s'], d.get('verified_by_oracle'))"
