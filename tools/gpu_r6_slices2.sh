#!/bin/bash
# round 6: does a SLICED program find its operands in the caches?  The round-5 assembly of the EVM-style constraint system (ZK_QUOTIENT_DAG=0: 109 k instructions, a fold per
# constraint, nothing parked across terms) can be cut anywhere: loop tool by slice count; then the sliced-program parity tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6slices2; mkdir -p $O
for sl in 0 4 16 64 0; do echo "== DAG=0, slices $sl"; ZK_QUOTIENT_DAG=0 ZK_QUOTIENT_TRACE=1 ZK_QUOTIENT_SLICES=$sl timeout 200 python tools/quot_evm_loop.py 20 3 2>&1 | grep -E "per launch|slices" | tail -2 | cut -c1-400; done
timeout 600 python -X faulthandler -m pytest tests/test_gpu_quotient.py -q -m gpu -x > $O/pytest_quot.log 2>&1; tail -5 $O/pytest_quot.log | cut -c1-200
