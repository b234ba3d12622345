#!/bin/bash
# round 6: sharded sessions after the chunked device exchange of the advice columns (ZK_SHARD_EXCHANGE_GROUPS)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_gpu_sharded_proof.py tests/test_gpu_bench_sharded.py tests/test_gpu_rccl_multirank.py -q -m gpu -x 2>&1 | tail -4
