#!/bin/bash
# round 6: same-box A/B of the interpreter variants: 16-word records + two register sets (default build), 4-word records + compiler-managed look-ahead (lib _q4), round 5's kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
L=$(pwd)/zkevm-circuits_amd/lib
for r in 1 2; do
for v in "" _q4; do echo "== lib$v"; ZKMI355_LIB=$L/libzkmi355$v.so timeout 120 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1; done
done
echo "== kernel 1"; ZK_QUOTIENT_KERNEL=1 timeout 120 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1
for v in "" _q4; do echo "== lib$v, every operand from ONE column"; ZKMI355_LIB=$L/libzkmi355$v.so timeout 120 python tools/quot_evm_loop.py 20 4 1 2>&1 | tail -1; done
echo "== kernel 1, ONE column"; ZK_QUOTIENT_KERNEL=1 timeout 120 python tools/quot_evm_loop.py 20 4 1 2>&1 | tail -1
for v in "" _q4; do echo "== lib$v quot_loop"; ZKMI355_LIB=$L/libzkmi355$v.so timeout 120 python tools/quot_loop.py 20 100 3 2>&1 | tail -1; done
timeout 600 python -m pytest tests/test_gpu_quotient.py tests/test_gpu_evm_shape.py -q -m gpu -x -k "not k20" 2>&1 | tail -3
