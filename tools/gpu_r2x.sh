#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2x; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench$i.json 2> $O/bench$i.err
python -c "
import json; d=json.load(open('$O/bench$i.json')); print(d['value'], d['ms_per_step'], d['extra']['kernel_avg_ms'], 'lone', d['extra']['msm_lone_ms'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
timeout 600 python bench_proof.py --k 24 --large --groups 3 --shplonk --pinned --repeat 2 > $O/k24.json 2> $O/k24.err; python -c "
import json; d=json.load(open('$O/k24.json')); print('k24', d['advice'], d['create_proof_s'], d['verified_by_oracle'], d['msm_count'])"
timeout 900 python bench_proof.py --k 25 --large --groups 2 --shplonk --pinned --repeat 2 > $O/k25.json 2> $O/k25.err; python -c "
import json; d=json.load(open('$O/k25.json')); print('k25', d['advice'], d['create_proof_s'], d['verified_by_oracle'], d['msm_count'])"; tail -2 $O/k25.err
