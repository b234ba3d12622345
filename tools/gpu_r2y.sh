#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2y; mkdir -p $O
export TMPDIR=/tmp
for h in 0 1 2; do
ZK_PERM_HINT=$h ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify > $O/sc$h.json 2> $O/sc$h.log
echo "SC hint $h: $(grep 'perm: commits' $O/sc$h.log | tail -1)"
ZK_PERM_HINT=$h ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 22 --large --groups 3 --shplonk --pinned --repeat 2 --no-verify > $O/rec$h.json 2> $O/rec$h.log
echo "rec hint $h: $(grep 'perm: commits' $O/rec$h.log | tail -1)"
done
