#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for gp in 4 6 8; do
  ZK_MSM_GRAPH_PIPES=$gp timeout 200 python tools/msm_graph_pipes.py 18 2>&1 | tail -4
  ZK_MSM_GRAPH_PIPES=$gp timeout 300 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 4 --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('keccak k18 pipes=$gp', d['create_proof_s'])"
done
ZK_MSM_GRAPH_PIPES=8 timeout 600 python -m pytest tests/test_gpu_msm.py -x -q -m gpu 2>&1 | tail -2
