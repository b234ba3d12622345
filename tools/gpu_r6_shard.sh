#!/bin/bash
# round 6: sharded sessions after the pair exchange moved to the device gather; the emulated rank; the bench line with projections
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/${1:-r6shard}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sharded_proof.py tests/test_gpu_bench_sharded.py tests/test_gpu_comm.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 1500 python bench.py --no-cpu-baseline --no-proof --no-msm-ntt --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
e = d["extra"]
print("value", d["value"], "projected", json.dumps(e["projected_rank_device_s"])[:800])
print("degree_blind", e["degree_blind"].get("value"), "structure_blind", e["structure_blind"].get("value"), "pcie", (e["pcie_inclusive"] or {}).get("value"))
PY
tail -5 $O/bench.err
