#!/bin/bash
# stream creation order vs hardware-queue sharing: advice-phase upload rate and the bench line per layout
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2ai; mkdir -p $O
export TMPDIR=/tmp
for lay in "$@"; do
ZK_STREAM_LAYOUT=$lay ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify > $O/sc.json 2> $O/sc.log
echo "layout $lay: $(grep 'advice upload' $O/sc.log | tail -1 | awk '{print $(NF-1)}') ms advice, proof $(python -c "import json; print(json.load(open('$O/sc.json'))['create_proof_s'][-1])")"
ZK_STREAM_LAYOUT=$lay timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('   bench', d['value'], d['ms_per_step'], 'lone', d['extra']['msm_lone_ms'])"
done
