#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3v; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_proof.py tests/test_gpu_quotient.py tests/test_gpu_sharded_proof.py -x -q -m gpu 2>&1 | tail -1
for v in "ZK_ADVICE_COSET_LATE_GB=0" "ZK_ADVICE_COSET_LATE_GB=48"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v ZK_PROVER_TRACE=1 timeout 600 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 4 --no-verify > $O/sc_$tag.json 2> $O/sc_$tag.err; echo "$v rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/sc_$tag.json").read().strip().splitlines()[-1])
print("create_proof_s",d["create_proof_s"])
PY
  grep "beside\|computed ahead" $O/sc_$tag.err | tail -2
  grep "zk prover" $O/sc_$tag.err | grep -v "quotient: \|plan\|ahead\|beside" | tail -16 | head -11
  grep "quotient: cosets" $O/sc_$tag.err | tail -8 | awk '{a+=$(NF-1)} END {print "  cosets of the columns (last proof):", a}'
  env $v timeout 600 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 4 --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('keccak k18', d['create_proof_s'])"
done
