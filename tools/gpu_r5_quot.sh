#!/bin/bash
# round 5: class programs as one weighted sum (ZK_QUOTIENT_GROUP) and the cheaper single-tuple lookup identity (ZK_LOOKUP_PLAIN=1 = halo2's form)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/${1:-r5quot}; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_proof.py tests/test_gpu_quotient.py tests/test_gpu_reference_protocol.py tests/test_gpu_mock.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
run() {
  local name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-proof --no-msm-ntt --steps 3 --warmup 1 > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]; e = d["extra"]
    print(sys.argv[2], "value", d["value"], "verified", e.get("verified_by_oracle"), "ntt_ms", r["device_ms_per_proof"], "transforms", r["transforms_per_proof"],
          "classes", e["kernel_class_device_ms_per_proof"], "blind", (e.get("structure_blind") or {}).get("value"))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
shift
for v in "$@"; do
  case $v in
    base) run base ZK_QUOTIENT_GROUP=0 ZK_LOOKUP_PLAIN=1 ;;
    lk) run lk ZK_QUOTIENT_GROUP=0 ;;
    group) run group ZK_X=0 ;;
    trace) run trace ZK_PROVER_TRACE=1 ZK_QUOTIENT_TRACE=1 ;;
  esac
done
