#!/bin/bash
# round 6: six waves per SIMD forced on the accumulator-in-memory evaluator (80 registers, 2 spilled); the plain shape with the compiler on / off
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
L=$(pwd)/zkevm-circuits_amd/lib
for v in "" _w6 "" _w6; do echo "== lib$v"; ZKMI355_LIB=$L/libzkmi355$v.so timeout 300 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1; done
bash tools/gpu_ab.sh ${1:-r6quot6}/plain 3 2 "ZK_BENCH_SHAPE=plain" "ZK_BENCH_SHAPE=plain ZK_QUOTIENT_DAG=0" "ZK_BENCH_SHAPE=plain ZK_QUOTIENT_ACC_MEM=0"
