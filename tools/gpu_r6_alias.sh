#!/bin/bash
# round 6: what a sliced class program would gain from the L2 -- 2^a consecutive workgroups of an XCD made to evaluate the SAME row tile (ZK_QUOTIENT_TILE_ALIAS=a, results wrong):
# the resident waves' operand working set shrinks by 2^a, the work per wave is unchanged
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for a in 0 2 3 4 5 6 7 0; do echo "== tile alias 2^$a"; ZK_QUOTIENT_TILE_ALIAS=$a timeout 120 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1; done
echo "== ONE column"; timeout 120 python tools/quot_evm_loop.py 20 4 1 2>&1 | tail -1
