#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
B=$PWD/zkevm-circuits_amd/lib
timeout 600 python -m pytest tests/test_gpu_msm.py -x -q -m gpu 2>&1 | tail -2
for lib in libzkmi355.so libzkmi355_tmin0.so libzkmi355.so libzkmi355_tmin0.so; do
  echo "== $lib"
  for k in 16 17 18; do ZKMI355_LIB=$B/$lib timeout 200 python tools/msm_graph_pipes.py $k 2>&1 | grep "of 16.*lagrange"; done
  ZKMI355_LIB=$B/$lib timeout 300 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 4 --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('keccak k18', d['create_proof_s'])"
  ZKMI355_LIB=$B/$lib timeout 300 python bench.py --no-proof --no-cpu-baseline --steps 32 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bench', d['value'], d['extra'].get('msm_lone_ms'))"
done
