#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
ZK_MSM_TRACE=1 timeout 300 python tools/msm_small_k.py 18 2>&1 | grep "zk msm\|hint=" | tail -12
