#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2r; mkdir -p $O
export TMPDIR=/tmp
for sel in supercircuit_shape_k20 keccak_shape_k18,supercircuit_shape_k20 recursion_shape_k22,supercircuit_shape_k20; do
  ZK_BENCH_PROOFS=$sel ZK_PROVER_TRACE=1 timeout 600 python bench.py --no-cpu-baseline --steps 4 --warmup 2 > $O/b.json 2> $O/b.err
  echo "== $sel"; grep "advice upload" $O/b.err | tail -3
done
