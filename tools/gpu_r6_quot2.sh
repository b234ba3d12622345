#!/bin/bash
# round 6: where the evaluator's time goes on the EVM-style headline -- operands from the caches (ALIAS), workgroups per CU (LDS pad)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
bash tools/gpu_ab.sh ${1:-r6quot2}/ab 2 1 "-" "ZK_QUOTIENT_ALIAS=1" "ZK_QUOTIENT_ALIAS=32" "ZK_QUOTIENT_LDS_PAD=40000" "ZK_QUOTIENT_LDS_PAD=53000" "ZK_QUOTIENT_LDS_PAD=80000" "ZK_QUOTIENT_LDS_PAD=147000"
