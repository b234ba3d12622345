#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2ah; mkdir -p $O
export TMPDIR=/tmp
for q in 2 3 4 5; do
GPU_MAX_HW_QUEUES=$q ZK_COPY_STREAMS=1 ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify > $O/sc.json 2> $O/sc.log
echo "hw queues $q: $(grep 'advice upload' $O/sc.log | tail -1) $(python -c "import json; print(json.load(open('$O/sc.json'))['create_proof_s'])")"
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('   bench', d['value'], d['ms_per_step'], 'lone', d['extra']['msm_lone_ms'])"
done
