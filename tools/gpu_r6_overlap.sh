#!/bin/bash
# round 6: coset transforms on the auxiliary stream beside the class programs -- parity, then A/B on one box, both shapes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/${1:-r6overlap}; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_proof.py tests/test_gpu_evm_shape.py tests/test_gpu_sharded_proof.py tests/test_gpu_headline_config.py tests/test_gpu_reference_protocol.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash tools/gpu_ab.sh ${1:-r6overlap}/evm 2 2 "-" "ZK_QUOTIENT_OVERLAP=0"
bash tools/gpu_ab.sh ${1:-r6overlap}/plain 3 2 "ZK_BENCH_SHAPE=plain" "ZK_BENCH_SHAPE=plain ZK_QUOTIENT_OVERLAP=0"
