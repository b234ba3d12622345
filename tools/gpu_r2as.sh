#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for cfg in "12 12" "11 12" "10 12" "11 11"; do
set -- $cfg
echo "last-pass log tile $1, other passes $2"; ZK_NTT_LAST_LOGTILE=$1 ZK_NTT_PASS_LOGTILE=$2 timeout 300 python tools/ntt_sizes.py 2>&1 | grep "k=20\|k=22\|k=24" | cut -c1-90
done
