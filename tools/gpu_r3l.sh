#!/bin/bash
# round 3, pass l: two MSM pipelines during the advice phase of the SuperCircuit shape?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3l; mkdir -p $O
export TMPDIR=/tmp
for v in "ZK_MSM_PIPES=1" "ZK_MSM_PIPES=2"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v ZK_PROVER_TRACE=1 timeout 600 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 4 --no-verify > $O/sc_$tag.json 2> $O/sc_$tag.err; echo "$v rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/sc_$tag.json").read().strip().splitlines()[-1])
print("create_proof_s",d["create_proof_s"])
PY
  grep "advice upload\|lookup phi\|perm: commits\|lookup m" $O/sc_$tag.err | tail -4
done
