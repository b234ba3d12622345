#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2h; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=12 --tb=short -p no:cacheprovider ) > $O/pytest.log 2>&1
grep -n "passed\|failed" $O/pytest.log | tail -3
timeout 200 python bench.py --no-proof --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json")); e=d["extra"]
print(d["value"], "Mscalar/s", d["ms_per_step"], "ms/step lone", e["msm_lone_ms"], e["kernel_avg_ms"])
PY
for split in 1 0; do
ZK_QUOTIENT_SPLIT=$split ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 > $O/sc_$split.json 2> $O/sc_trace_$split.log
python -c "
import json; d=json.load(open('$O/sc_$split.json')); print('split=$split supercircuit shape', d['create_proof_s'], 'verified', d['verified_by_oracle'])"
done
grep "quotient" $O/sc_trace_1.log | tail -18
ZK_PROVER_TRACE=1 timeout 300 python bench_proof.py --k 18 --keccak --shplonk --pinned --repeat 2 > $O/kc.json 2> $O/kc_trace.log
python -c "
import json; d=json.load(open('$O/kc.json')); print('keccak shape', d['create_proof_s'], 'verified', d['verified_by_oracle'])"
