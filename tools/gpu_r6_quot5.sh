#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
L=$(pwd)/zkevm-circuits_amd/lib
for v in "" _prev "" _prev; do echo "== lib$v"; ZKMI355_LIB=$L/libzkmi355$v.so timeout 300 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1; done
timeout 600 python -m pytest tests/test_gpu_quotient.py tests/test_gpu_evm_shape.py -q -m gpu -x -k "not k20" 2>&1 | tail -2
