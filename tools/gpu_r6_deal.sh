#!/bin/bash
# round 6: (class, coset) pairs dealt by marginal cost -- sharded parity, then every rank of N = 2 / 4 / 8 emulated under both deals
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6deal; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_sharded_proof.py tests/test_gpu_bench_sharded.py -q -m gpu -x 2>&1 | tail -3
for d in 1 0; do
ZK_SHARD_DEAL=$d timeout 1500 python bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify --steps 1 --warmup 1 > $O/bench_deal$d.json 2> $O/bench_deal$d.err; echo "deal $d rc=$? t=$SECONDS"
python - $O/bench_deal$d.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
p = d["extra"]["projected_rank_device_s"]
print(d["value"], p.get("rank_device_s"), p.get("by_rank_s"), p.get("error"))
PY
done
