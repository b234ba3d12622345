#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2an; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_proof.py tests/test_gpu_sharded_proof.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|Error" $O/pytest.log | tail -3
for pipes in 2 1; do
ZK_MSM_PIPES=$pipes ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 4 --no-verify > $O/kc.json 2> $O/kc.log
echo "pipes=$pipes keccak $(python -c "import json; print(json.load(open('$O/kc.json'))['create_proof_s'])") $(grep 'advice upload\|h commits\|lookup phi ' $O/kc.log | tail -3 | awk '{printf "%s %s | ", $(NF-3), $(NF-1)}')"
done
timeout 300 python tools/msm_small_k.py 18 2>&1 | grep "hint=" | head -4
ZK_MSM_PIPES=2 timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('k=20 two pipes', d['value'], d['ms_per_step'])"
