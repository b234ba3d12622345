#!/bin/bash
# round 6: the interpreter with fixed register roles (k_quotient_eval2) against round 5's kernel (ZK_QUOTIENT_KERNEL=1) on the EVM-style class program;
# LDS slots below the top 2 (default; deeper entries in memory) / 3, then the evaluator's parity tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
L=$(pwd)/zkevm-circuits_amd/lib
for v in 2 1 2; do echo "== kernel $v"; ZK_QUOTIENT_KERNEL=$v timeout 300 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1; done
echo "== kernel 2, 3 LDS slots"; ZK_QUOTIENT_LDS_SLOTS=3 timeout 300 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1
echo "== kernel 2, 1 LDS slot"; ZK_QUOTIENT_LDS_SLOTS=1 timeout 300 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1
for v in 2 1; do echo "== kernel $v, quot_loop"; ZK_QUOTIENT_KERNEL=$v timeout 300 python tools/quot_loop.py 20 100 3 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_gpu_quotient.py tests/test_gpu_evm_shape.py tests/test_gpu_proof.py -q -m gpu -x -k "not k20" 2>&1 | tail -3
ZK_QUOTIENT_LDS_SLOTS=1 timeout 900 python -m pytest tests/test_gpu_quotient.py tests/test_gpu_evm_shape.py -q -m gpu -x -k "not k20" 2>&1 | tail -3
