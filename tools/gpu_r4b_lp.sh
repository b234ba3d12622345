#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r4b_lp; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_proof.py tests/test_gpu_baseline_sizes.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$? t=${SECONDS}"; tail -2 $O/pytest.log
for v in 1 2; do
  ZK_PROVER_TRACE=1 timeout 400 python bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify 2> $O/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['extra'].get('pcie_inclusive',{}).get('value'))"
  grep "advice upload + commits" $O/err_$v.log | sed -n 4,6p | tr '\n' ' '; echo
done
timeout 200 python tools/msm_dist.py 20 64 survey_60_30_10,small16 1 2>&1 | grep "k=20"
echo "t=${SECONDS}"
