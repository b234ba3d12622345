#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 300 python bench.py --proof-worker evm_shape_k14_mock 2>&1 | tail -3
timeout 600 python bench.py --proof-worker supercircuit_shape_k20_mock 2>&1 | tail -3
