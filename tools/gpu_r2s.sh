#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2s; mkdir -p $O
export TMPDIR=/tmp
for pre in torch_msm msm_prof msm_ntt msm_lone torch_msm_prof_ntt_lone; do
  ZK_PROVER_TRACE=1 timeout 300 python tools/upload_order.py $pre > $O/u_$pre.out 2> $O/u_$pre.err
  tail -1 $O/u_$pre.out; grep "advice upload" $O/u_$pre.err | tail -1
done
