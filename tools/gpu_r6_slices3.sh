#!/bin/bash
# round 6: slice count in the whole proof
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
bash tools/gpu_ab.sh r6slices3/evm 3 1 "-" "ZK_QUOTIENT_SLICES=24" "ZK_QUOTIENT_SLICES=48" "ZK_QUOTIENT_SLICES=81"
ZK_QUOTIENT_TRACE=1 ZK_BENCH_QUICK=1 timeout 600 python bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify --steps 1 --warmup 1 2>&1 >/dev/null | grep " slices" | sort | uniq -c | sort -rn | head -6 | cut -c1-900
bash tools/gpu_ab.sh r6slices3/plain 3 1 "ZK_BENCH_SHAPE=plain" "ZK_BENCH_SHAPE=plain ZK_QUOTIENT_SLICES=0"
