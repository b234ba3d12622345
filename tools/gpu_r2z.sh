#!/bin/bash
# round-2 record run: GPU test suite, smoke, the driver's bench line, kernel statistics and PMC traffic of the bench command, kernel statistics of the SuperCircuit-shape proof
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2z; mkdir -p $O
export TMPDIR=/tmp
run_bounded() { local secs=$1 log=$2; shift 2; setsid "$@" > "$log" 2>&1 & local pid=$!; ( sleep "$secs"; kill -TERM -- -"$pid" 2>/dev/null; sleep 3; kill -KILL -- -"$pid" 2>/dev/null ) & local wd=$!; wait "$pid"; local rc=$?; kill "$wd" 2>/dev/null; return $rc; }
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 900 python bench.py ) > $O/bench_full.json 2> $O/bench_full.err
python - <<PY
import json
d=json.load(open("$O/bench_full.json"))
print(d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
for k,v in d.get("proof",{}).items(): print(k, {x:v.get(x) for x in ("value","create_proof_s","verified_by_oracle","error","chain_of_4_proofs_s")}, (v.get("roofline_quotient") or {}).get("frac"))
print(d.get("cpu_baseline"))
PY
ZK_BENCH_DENSE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 > $O/sc_dense.json 2> $O/sc_dense.log
python -c "
import json; d=json.load(open('$O/sc_dense.json')); print('dense witness', d['create_proof_s'], d['verified_by_oracle'])"
cd /tmp
run_bounded 150 $O/prof_bench.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $ROOT/bench.py --no-proof --no-cpu-baseline
echo "kernel trace rc=$?"
run_bounded 150 $O/pmc_fetch.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $ROOT/bench.py --no-proof --no-cpu-baseline --steps 16 --warmup 8
echo "pmc fetch rc=$?"
run_bounded 150 $O/pmc_write.log rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $ROOT/bench.py --no-proof --no-cpu-baseline --steps 16 --warmup 8
echo "pmc write rc=$?"
run_bounded 200 $O/prof_sc.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sc -- python $ROOT/bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify
echo "sc trace rc=$?"
