#!/bin/bash
# round 6: the +beta additions of the logUp sums as one batched launch -- parity (proof tests), then the stage table
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r6phi; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_proof.py tests/test_gpu_evm_shape.py tests/test_gpu_headline_config.py tests/test_gpu_reference_protocol.py tests/test_gpu_sharded_proof.py -q -m gpu -x 2>&1 | tail -3
for shape in evm plain; do
ZK_BENCH_SHAPE=$shape ZK_PROVER_TRACE=1 ZK_BENCH_QUICK=1 timeout 900 python bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify --steps 2 --warmup 1 > $O/$shape.json 2> $O/$shape.err
python tools/trace_stages.py $O/$shape.err 2 2>/dev/null | grep -E "lookup|sum of the marks|quotient eval"
done
