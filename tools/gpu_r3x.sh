#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_proof.py -x -q -m gpu -k "groups_of_small or cosets_computed_ahead" 2>&1 | tail -3
