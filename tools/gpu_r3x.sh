#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cpp_caller.py -x -q -m gpu 2>&1 | tail -15
