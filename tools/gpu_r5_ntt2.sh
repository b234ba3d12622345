#!/bin/bash
# round 5: tile sizes of the fixed-structure passes (workgroups per CU), standalone, then the headline proof for the chosen ones
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/${1:-r5ntt2}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ntt.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for cfg in "12 11 0" "11 11 0" "10 11 0" "12 10 1" "11 10 1" "11 11 1" "10 10 1" "12 12 0"; do
  set -- $cfg
  ZK_NTT_PASS_LOGTILE=$1 ZK_NTT_LAST_LOGTILE=$2 ZK_NTT_XCD=$3 timeout 120 python tools/ntt_batch_time.py 20 32 10 2>&1 | grep "us per" | grep -v "x4" | sed "s/^/pass=$1 last=$2 xcd=$3 /"
done | tee $O/tiles.txt
ZK_NTT_PASS_LOGTILE=11 timeout 300 python -m pytest tests/test_gpu_ntt.py -q -m gpu -x 2>&1 | tail -1
ZK_NTT_PASS_LOGTILE=10 ZK_NTT_LAST_LOGTILE=10 ZK_NTT_XCD=1 timeout 300 python -m pytest tests/test_gpu_ntt.py -q -m gpu -x 2>&1 | tail -1
