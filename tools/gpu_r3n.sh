#!/bin/bash
# round 3, pass n: degree classes with shared intermediates (CSE keys) and in sharded sessions -- tests, SuperCircuit shape with CSE
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3n; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_proof.py tests/test_gpu_quotient.py tests/test_gpu_sharded_proof.py tests/test_gpu_comm.py tests/test_gpu_rccl_multirank.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" $O/pytest.log | tail -3
for v in "ZK_BENCH_CSE=1" "ZK_BENCH_CSE=1 ZK_QUOTIENT_SPLIT=0"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v ZK_PROVER_TRACE=1 timeout 600 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 > $O/sc_$tag.json 2> $O/sc_$tag.err; echo "$v rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/sc_$tag.json").read().strip().splitlines()[-1])
print("create_proof_s",d["create_proof_s"],"verified",d.get("verified_by_oracle"))
PY
  grep "quotient: program" $O/sc_$tag.err | tail -8 | awk '{a+=$(NF-1)} END {print "  programs (last proof):", a}'
  grep "quotient: cosets" $O/sc_$tag.err | tail -8 | awk '{a+=$(NF-1)} END {print "  cosets of the columns (last proof):", a}'
done
