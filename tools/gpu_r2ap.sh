#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2ap; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_proof.py tests/test_gpu_sharded_proof.py tests/test_gpu_api_edges.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|Error" $O/pytest.log | tail -3
for g in 1 0; do
ZK_MSM_GRAPH=$g ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 4 > $O/kc$g.json 2> $O/kc$g.log
echo "graph=$g keccak $(python -c "import json; d=json.load(open('$O/kc$g.json')); print(d['create_proof_s'], d['verified_by_oracle'])") $(grep 'advice upload\|h commits\|lookup phi \|multiopen' $O/kc$g.log | tail -4 | awk '{printf "%s %s | ", $(NF-3), $(NF-1)}')"
done
ZK_MSM_TRACE=1 timeout 300 python tools/msm_small_k.py 18 2>&1 | grep "zk msm" | tail -4
