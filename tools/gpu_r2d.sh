#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_proof.py tests/test_gpu_field.py -q --maxfail=12 --tb=short -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for cfg in "0 x" "30 x" "30 1" "0 1"; do
  set -- $cfg; bits=$1; nar=$2
  if [ "$nar" = "x" ]; then unset ZK_MSM_NARROW; else export ZK_MSM_NARROW=$nar; fi
  ZK_BENCH_SCALAR_BITS=$bits timeout 200 python bench.py --no-proof --no-cpu-baseline > $O/bench_b${bits}_n$nar.json 2> $O/bench_b${bits}_n$nar.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_b${bits}_n$nar.json")); e=d["extra"]
    print("bits=$bits narrow=$nar", d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", e["kernel_avg_ms"])
except Exception as ex: print("bits=$bits narrow=$nar FAILED", ex)
PY
done
unset ZK_MSM_NARROW
ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify > $O/sc.json 2> $O/sc_trace.log
python -c "
import json; d=json.load(open('$O/sc.json')); print('supercircuit shape', d['create_proof_s'], 'keygen', d['keygen_pk_s'])"
grep -v "quotient:" $O/sc_trace.log | tail -16
ZK_PROVER_TRACE=1 timeout 300 python bench_proof.py --k 18 --keccak --shplonk --pinned --repeat 2 > $O/kc.json 2> $O/kc_trace.log
python -c "
import json; d=json.load(open('$O/kc.json')); print('keccak shape', d['create_proof_s'], 'verified', d['verified_by_oracle'])"
grep -v "quotient:" $O/kc_trace.log | tail -16
