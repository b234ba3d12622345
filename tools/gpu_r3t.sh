#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3t; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_proof.py -x -q -m gpu 2>&1 | tail -1
for a in "20 30" "20 16" "20 8" "18 8"; do timeout 120 python tools/msm_narrow.py $a 32 2>&1 | grep "hint=1"; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-proof --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read())
print("value",d["value"],"ms/step",d["ms_per_step"],d["extra"]["kernel_avg_ms"])
PY
ZK_PROVER_TRACE=1 timeout 600 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 4 --no-verify > $O/sc.json 2> $O/sc.err
python - <<PY
import json
d=json.loads(open("$O/sc.json").read().strip().splitlines()[-1])
print("create_proof_s",d["create_proof_s"])
PY
grep "advice upload" $O/sc.err | tail -2
ZK_PROVER_TRACE=1 timeout 600 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 4 --no-verify > $O/kc.json 2> $O/kc.err
python - <<PY
import json
d=json.loads(open("$O/kc.json").read().strip().splitlines()[-1])
print("keccak create_proof_s",d["create_proof_s"])
PY
