#!/bin/bash
# pass J: one-launch partition sort of the merged MSM, lowered expression evaluator (memory operands, lazy sums, prefetch)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2j; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_quotient.py tests/test_gpu_proof.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for bs in 1 0; do
  ZK_MSM_BINSORT=$bs timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench_bs$bs.json 2> $O/bench_bs$bs.err
  python - <<PY
import json
d=json.load(open("$O/bench_bs$bs.json"))
print("binsort=$bs", d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
PY
done
for fuse in 1 0; do
  ZK_QUOTIENT_FUSE=$fuse ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 > $O/sc_fuse$fuse.json 2> $O/sc_fuse$fuse.log
  python -c "
import json; d=json.load(open('$O/sc_fuse$fuse.json')); print('fuse=$fuse supercircuit shape', d['create_proof_s'], 'verified', d['verified_by_oracle'])"
  grep "quotient: program" $O/sc_fuse$fuse.log | tail -8 | awk '{s+=$(NF-1)} END {print "  program ms per proof:", s}'
done
ZK_QUOTIENT_SPLIT=0 ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify > $O/sc_nosplit.json 2> $O/sc_nosplit.log
grep "quotient: program" $O/sc_nosplit.log | tail -8
timeout 300 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 3 > $O/kc.json 2> $O/kc.log; python -c "
import json; d=json.load(open('$O/kc.json')); print('keccak shape', d['create_proof_s'], d['verified_by_oracle'])"
