#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sharded_proof.py -q -m gpu -x -k "knobs10" 2>&1 | grep -v "^$" | tail -40 | cut -c1-300
