#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 300 python tools/ntt_sizes.py 2>&1 | tail -7
