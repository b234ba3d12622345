#!/bin/bash
# round 6, first pass: the new parity tests (EVM-style shape, plans), then the two headline shapes under the stage trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/${1:-r6first}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_evm_shape.py tests/test_gpu_quotient.py tests/test_gpu_proof.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for shape in evm plain; do
  ZK_BENCH_SHAPE=$shape ZK_PROVER_TRACE=1 ZK_QUOTIENT_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --no-proof --no-msm-ntt --steps 2 --warmup 1 > $O/$shape.json 2> $O/$shape.err
  echo "== $shape rc=$?"
  python tools/trace_stages.py $O/$shape.err 2>/dev/null | head -60
  grep "zk quotient\] plan" $O/$shape.err | sort | uniq -c | head
  python - "$O/$shape.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e = d["extra"]
    print("value", d["value"], "verified", e["verified_by_oracle"], "classes", e["kernel_class_device_ms_per_proof"])
    print("structure_blind", e["structure_blind"].get("value"), "degree_blind", {k: v for k, v in e["degree_blind"].items() if k != "plan" and k != "note"}, "pcie", (e["pcie_inclusive"] or {}).get("value"))
    print("evaluator", json.dumps(e["evaluator"])[:1500])
    print("roofline", d["roofline"]["frac"], d["roofline"]["transforms_per_proof"], d["roofline"]["avg_launch_ms"])
except Exception as ex:
    print("FAILED", ex)
PY
done
