#!/bin/bash
# round 5: the fixed-structure NTT passes (ZK_NTT_FIXED) -- parity tests, standalone batch timings, headline proof with the knob off / on
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/${1:-r5ntt}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_field.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for f in 0 1; do
  for kk in 20 18 21; do ZK_NTT_FIXED=$f timeout 120 python tools/ntt_batch_time.py $kk 32 10 2>&1 | grep "us per" | sed "s/^/fixed=$f /"; done
done | tee $O/batch_time.txt
run() {
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
    fixed0) run fixed0 ZK_NTT_FIXED=0 ;;
    fixed1) run fixed1 ZK_NTT_FIXED=1 ;;
  esac
done
