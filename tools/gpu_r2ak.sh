#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2ak; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_api_edges.py tests/test_gpu_comm.py -x -q -m gpu > $O/b.log 2>&1; echo "edges+comm rc=$?"; grep -n "passed\|failed" $O/b.log | tail -1
timeout 600 python -m pytest tests/test_gpu_comm.py -x -q -m gpu > $O/c.log 2>&1; echo "comm rc=$?"; grep -n "passed\|failed" $O/c.log | tail -1
timeout 900 python -m pytest tests/test_gpu_sharded_ntt.py tests/test_gpu_comm.py tests/test_gpu_sharded_proof.py -x -q -m gpu > $O/d.log 2>&1; echo "sharded+comm rc=$?"; grep -n "passed\|failed" $O/d.log | tail -1
