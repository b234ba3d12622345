#!/bin/bash
# round 6: the evaluator loop on the compiled EVM-style program under library variants, then its issue counters
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
L=$(pwd)/zkevm-circuits_amd/lib
for v in "" _base _pf2 _w7 _w8; do echo "== lib$v"; ZKMI355_LIB=$L/libzkmi355$v.so timeout 300 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1; done
echo "== relaxed off"; ZK_QUOTIENT_RELAXED=0 timeout 300 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1
echo "== dag off"; ZK_QUOTIENT_DAG=0 timeout 300 python tools/quot_evm_loop.py 20 2 2>&1 | tail -1
for nb in 1 8 64; do echo "== $nb buffers"; timeout 300 python tools/quot_evm_loop.py 20 4 $nb 2>&1 | tail -1; done
bash tools/pmc_quot_evm.sh ${1:-r6quot3}/pmc
