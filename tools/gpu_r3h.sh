#!/bin/bash
# round 3, pass h: cosets of the advice columns computed during the advice phase -- proof tests, SuperCircuit-shape A/B with stage trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_proof.py tests/test_gpu_quotient.py tests/test_gpu_sharded_proof.py tests/test_gpu_comm.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for v in 0 64; do
  ZK_ADVICE_COSET_GB=$v ZK_PROVER_TRACE=1 timeout 600 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 > $O/sc_$v.json 2> $O/sc_$v.err; echo "sc cosets=$v rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/sc_$v.json").read().strip().splitlines()[-1])
print("create_proof_s",d["create_proof_s"],"verified",d.get("verified_by_oracle"))
PY
  grep "zk prover" $O/sc_$v.err | grep -v "quotient: " | tail -22
  grep "quotient: " $O/sc_$v.err | tail -16 | awk '{a[$4" "$5" "$6]+=$(NF-1)} END {for (k in a) print "  sum", k, a[k]}'
done
rocm-smi --showmeminfo vram 2>/dev/null | tail -3
