#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mock.py -x -q -m gpu 2>&1 | tail -30
