#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2l; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_quotient.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
ZK_QUOTIENT_FUSE=1 timeout 200 python tools/quot_loop.py 20 100 4 2>&1 | tail -1
ZK_QUOTIENT_FUSE=0 timeout 200 python tools/quot_loop.py 20 100 4 2>&1 | tail -1
for bs in 1 0; do
  ZK_MSM_BINSORT=$bs timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench_bs$bs.json 2> $O/bench_bs$bs.err
  python - <<PY
import json
d=json.load(open("$O/bench_bs$bs.json"))
print("binsort=$bs", d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
PY
done
