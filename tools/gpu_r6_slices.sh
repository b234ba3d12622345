#!/bin/bash
# round 6: class programs cut into slices that share their rows' operands through the caches (plan_slices; class compiler emitting term by term) -- loop tool by slice count,
# parity, whole proof A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for sl in 0 -1 4 8 32 64 0; do
  if [ $sl = -1 ]; then echo "== slices: default"; ZK_QUOTIENT_TRACE=1 timeout 120 python tools/quot_evm_loop.py 20 4 2>&1 | grep -E " slices|per launch" | tail -2 | cut -c1-700
  else echo "== slices $sl"; ZK_QUOTIENT_SLICES=$sl timeout 120 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1; fi
done
echo "== chunked emission off, one piece"; ZK_QUOTIENT_CHUNK=0 timeout 120 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_quotient.py -q -m gpu -x 2>&1 | tail -2
[ "$1" = "quick" ] && exit 0
bash tools/gpu_ab.sh ${1:-r6slices}/evm 3 1 "-" "ZK_QUOTIENT_SLICES=0" "ZK_QUOTIENT_SLICES=0 ZK_QUOTIENT_CHUNK=0"
timeout 1500 python -m pytest tests/test_gpu_evm_shape.py tests/test_gpu_proof.py -q -m gpu -x 2>&1 | tail -3
