#!/bin/bash
# round 6: k_quotient_eval2 (fixed register roles, 16-word records) -- loop tool A/B, parity, then the whole proof A/B on both shapes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for v in 2 1 2; do echo "== kernel $v"; ZK_QUOTIENT_KERNEL=$v timeout 120 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1; done
echo "== kernel 2, every operand from ONE column (loads from L2)"; timeout 120 python tools/quot_evm_loop.py 20 4 1 2>&1 | tail -1
echo "== kernel 2, 3 LDS slots"; ZK_QUOTIENT_LDS_SLOTS=3 timeout 120 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1
for v in 2 1; do echo "== kernel $v, quot_loop"; ZK_QUOTIENT_KERNEL=$v timeout 120 python tools/quot_loop.py 20 100 3 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_gpu_quotient.py tests/test_gpu_evm_shape.py tests/test_gpu_proof.py tests/test_gpu_mock.py -q -m gpu -x -k "not k20" 2>&1 | tail -3
ZK_QUOTIENT_LDS_SLOTS=1 timeout 300 python -m pytest tests/test_gpu_quotient.py tests/test_gpu_evm_shape.py -q -m gpu -x -k "not k20" 2>&1 | tail -3
[ "$1" = "quick" ] && exit 0
bash tools/gpu_ab.sh ${1:-r6eval2}/evm 3 1 "-" "ZK_QUOTIENT_KERNEL=1"
bash tools/gpu_ab.sh ${1:-r6eval2}/plain 3 1 "ZK_BENCH_SHAPE=plain" "ZK_BENCH_SHAPE=plain ZK_QUOTIENT_KERNEL=1" "ZK_BENCH_SHAPE=plain ZK_QUOTIENT_DAG=0"
