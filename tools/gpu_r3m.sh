#!/bin/bash
# round 3, pass m: is the advice phase bound by the uploads or by the device?  (uploads skipped = garbage witness, timing only)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3m; mkdir -p $O
export TMPDIR=/tmp
for v in "ZK_DEBUG_SKIP_UPLOAD=1 ZK_ADVICE_COSET_GB=64" "ZK_DEBUG_SKIP_UPLOAD=1 ZK_ADVICE_COSET_GB=0"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v ZK_PROVER_TRACE=1 timeout 600 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 --no-verify > $O/sc_$tag.json 2> $O/sc_$tag.err; echo "$v rc=$?"
  grep "advice upload" $O/sc_$tag.err | tail -3
  tail -2 $O/sc_$tag.err | cut -c1-200
done
