#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3p; mkdir -p $O
export TMPDIR=/tmp
ZK_PROVER_TRACE=1 ZK_BENCH_PROOFS=supercircuit_shape_k20 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err; echo "rc=$?"
grep "plan\|ahead\|advice upload" $O/b.err | head -8
python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
for k,v in d.get("proof",{}).items(): print(k, v.get("value"), v.get("create_proof_s"))
PY
rocm-smi --showmeminfo vram | grep -i "used"
