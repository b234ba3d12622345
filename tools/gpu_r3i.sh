#!/bin/bash
# round 3, pass i: why no cosets were planned; does the auxiliary stream's priority slow the uploads?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp
for v in "ZK_ADVICE_COSET_GB=64" "ZK_ADVICE_COSET_GB=40"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v ZK_PROVER_TRACE=1 timeout 600 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 --no-verify > $O/sc_$tag.json 2> $O/sc_$tag.err; echo "$v rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/sc_$tag.json").read().strip().splitlines()[-1])
print("create_proof_s",d["create_proof_s"])
PY
  grep "plan\|computed ahead\|advice upload" $O/sc_$tag.err | tail -4; grep "zk prover" $O/sc_$tag.err | grep -v "quotient: \|plan\|ahead" | tail -16
  grep "quotient: cosets" $O/sc_$tag.err | tail -8 | awk '{a+=$(NF-1)} END {print "  cosets of the columns (last proof):", a}'
done
