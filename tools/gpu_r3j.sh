#!/bin/bash
# round 3, pass j: inline combination for every position -- tests, bench, SuperCircuit shape (sparse and dense witness)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3j; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_proof.py tests/test_gpu_params.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | head -1
timeout 600 python bench.py --steps 48 --warmup 16 --no-proof --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read())
print("value",d["value"],"ms/step",d["ms_per_step"],d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
PY
for v in "ZK_X=0" "ZK_BENCH_DENSE=1"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v ZK_PROVER_TRACE=1 timeout 600 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 4 --no-verify > $O/sc_$tag.json 2> $O/sc_$tag.err; echo "$v rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/sc_$tag.json").read().strip().splitlines()[-1])
print("create_proof_s",d["create_proof_s"])
PY
  grep "computed ahead\|advice upload" $O/sc_$tag.err | tail -2
  grep "quotient: cosets" $O/sc_$tag.err | tail -8 | awk '{a+=$(NF-1)} END {print "  cosets of the columns (last proof):", a}'
done
