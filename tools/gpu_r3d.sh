#!/bin/bash
# round 3, pass d: small-grid combine kernels -- MSM tests, bench line, SuperCircuit-shape stage trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_proof.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 600 python bench.py --steps 48 --warmup 16 --no-proof --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read())
print("value",d["value"],"ms/step",d["ms_per_step"],d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
PY
ZK_PROVER_TRACE=1 timeout 600 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 > $O/sc.json 2> $O/sc.err; echo "sc rc=$?"
tail -c 1500 $O/sc.json; grep "zk prover" $O/sc.err | tail -45
