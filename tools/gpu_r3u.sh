#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3u; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py -x -q -m gpu 2>&1 | tail -1
for g in 2 4 8; do
  echo "ZK_MSM_NARROW_GROUP=$g"
  for a in "20 30" "20 8" "18 8"; do ZK_MSM_NARROW_GROUP=$g timeout 120 python tools/msm_narrow.py $a 32 2>&1 | grep "hint=1" | cut -c1-80; done
  ZK_MSM_NARROW_GROUP=$g timeout 600 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 4 --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('keccak k18', d['create_proof_s'])"
  ZK_MSM_NARROW_GROUP=$g timeout 600 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sc k20', d['create_proof_s'])"
done
