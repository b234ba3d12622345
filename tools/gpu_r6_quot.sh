#!/bin/bash
# round 6: the evaluator on the EVM-style headline -- parity tests of what changed, then library variants / knobs alternating on ONE box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/${1:-r6quot}; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_evm_shape.py tests/test_gpu_quotient.py tests/test_gpu_proof.py tests/test_gpu_field.py tests/test_gpu_ntt.py tests/test_gpu_baseline_sizes.py tests/test_gpu_mock.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
L=$ROOT/zkevm-circuits_amd/lib
bash tools/gpu_ab.sh ${1:-r6quot}/ab 2 2 "ZKMI355_LIB=$L/libzkmi355_base.so" "-" "ZKMI355_LIB=$L/libzkmi355_pf2.so" "ZK_QUOTIENT_ALIAS=1" "ZK_QUOTIENT_ALIAS=1 ZKMI355_LIB=$L/libzkmi355_base.so"
