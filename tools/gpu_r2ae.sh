#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 300 python tools/msm_runs.py 2000 2>&1 | tail -6
timeout 300 python tools/msm_runs.py 50000 2>&1 | grep runs
