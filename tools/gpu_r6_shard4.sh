#!/bin/bash
# round 6: where the emulated rank of N = 8 spends its time at the end of the round (stage trace of the bench's projection runs)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6shard4; mkdir -p $O
ZK_PROVER_TRACE=1 timeout 1500 python bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify --steps 1 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python tools/trace_stages.py $O/bench.err 1 2>/dev/null | head -60
