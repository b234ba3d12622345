#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2au; mkdir -p $O
export TMPDIR=/tmp
for cfg in "11 11" "12 12" "11 12"; do
set -- $cfg
ZK_NTT_LAST_LOGTILE=$1 ZK_NTT_PASS_LOGTILE=$2 ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 22 --large --groups 3 --shplonk --pinned --repeat 3 --no-verify > $O/rec.json 2> $O/rec.log
echo "last $1 pass $2: $(python -c "import json; print(json.load(open('$O/rec.json'))['create_proof_s'])")"
grep "zk prover" $O/rec.log | tail -33 | grep -v "quotient: program" | awk '{printf "%s=%s ", $(NF-3), $(NF-1)} END {print ""}'
done
