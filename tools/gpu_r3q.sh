#!/bin/bash
# round 3, pass q: XCD-aware tile numbering of the last NTT pass, A/B (same box) + the NTT tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3q; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu 2>&1 | tail -1
for v in 0 1 0 1; do echo "ZK_NTT_XCD=$v"; ZK_NTT_XCD=$v timeout 120 python tools/ntt_sizes.py 2>&1 | grep "k=18\|k=20\|k=22\|k=24"; done
