#!/bin/bash
# round-2 record run: the driver's bench line (with proofs), the dense-witness variant, kernel statistics of the proof
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2i; mkdir -p $O
export TMPDIR=/tmp
run_bounded() { local secs=$1 log=$2; shift 2; setsid "$@" > "$log" 2>&1 & local pid=$!; ( sleep "$secs"; kill -TERM -- -"$pid" 2>/dev/null; sleep 3; kill -KILL -- -"$pid" 2>/dev/null ) & local wd=$!; wait "$pid"; local rc=$?; kill "$wd" 2>/dev/null; return $rc; }
( time timeout 900 python bench.py ) > $O/bench_full.json 2> $O/bench_full.err
python - <<PY
import json
d=json.load(open("$O/bench_full.json"))
print(d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
for k,v in d.get("proof",{}).items(): print(k, {x:v.get(x) for x in ("value","create_proof_s","verified_by_oracle","error","chain_of_4_proofs_s")}, v.get("roofline_quotient"))
print(d.get("cpu_baseline"))
PY
ZK_BENCH_DENSE=1 ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 > $O/sc_dense.json 2> $O/sc_dense_trace.log
python -c "
import json; d=json.load(open('$O/sc_dense.json')); print('dense witness: supercircuit shape', d['create_proof_s'], 'verified', d['verified_by_oracle'])"
grep "advice upload" $O/sc_dense_trace.log | tail -1
ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify > $O/sc.json 2> $O/sc_trace.log
grep -v "quotient:" $O/sc_trace.log | tail -16
cd /tmp
run_bounded 200 $O/prof_sc.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sc -- python $ROOT/bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify
echo "sc trace rc=$?"
run_bounded 150 $O/prof_bench.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $ROOT/bench.py --no-proof --no-cpu-baseline
echo "bench trace rc=$?"
