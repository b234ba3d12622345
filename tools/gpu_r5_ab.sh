#!/bin/bash
# round 5: parity tests of what changed, then the headline proof under measurement knobs (one line per variant)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/${1:-r5ab}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_proof.py tests/test_gpu_ntt.py tests/test_gpu_msm.py tests/test_gpu_quotient.py tests/test_gpu_headline_config.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify --steps 3 --warmup 1 > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]; e = d["extra"]
    print(sys.argv[2], "value", d["value"], "ntt_ms", r["device_ms_per_proof"], "transforms", r["transforms_per_proof"], "us/transform", round(r["avg_launch_ms"] * 1e3, 1),
          "classes", e["kernel_class_device_ms_per_proof"], "blind", (e.get("structure_blind") or {}).get("value"))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
shift
for v in "$@"; do
  case $v in
    base) run base ZK_X=0 ;;
    nosplit) run nosplit ZK_QUOTIENT_ADDSPLIT=0 ;;
    noxcd) run noxcd ZK_NTT_XCD_COLS=0 ;;
    batch8) run batch8 ZK_NTT_BATCH=8 ;;
    batch16) run batch16 ZK_NTT_BATCH=16 ;;
    trace) run trace ZK_PROVER_TRACE=1 ;;
  esac
done
