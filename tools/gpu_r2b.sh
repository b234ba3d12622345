#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_proof.py tests/test_gpu_msm.py tests/test_gpu_api_edges.py -q --maxfail=12 --tb=short -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json")); e=d["extra"]
print(d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", e["kernel_avg_ms"])
PY
ZK_PROVER_TRACE=1 timeout 600 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify > $O/sc.json 2> $O/sc_trace.log
tail -c 400 $O/sc.json; grep -v "quotient:" $O/sc_trace.log | tail -32
