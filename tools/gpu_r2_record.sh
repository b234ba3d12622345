#!/bin/bash
# Record run of the shipped build on ONE box: the driver's bench line, then kernel statistics and PMC traffic of the same
# command, then kernel statistics of the SuperCircuit-shape proof (so that bench.py's own kernel averages and the
# rocprofv3 summaries under profiles/ come from the same machine).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2rec; mkdir -p $O
export TMPDIR=/tmp
run_bounded() { local secs=$1 log=$2; shift 2; setsid "$@" > "$log" 2>&1 & local pid=$!; ( sleep "$secs"; kill -TERM -- -"$pid" 2>/dev/null; sleep 3; kill -KILL -- -"$pid" 2>/dev/null ) & local wd=$!; wait "$pid"; local rc=$?; kill "$wd" 2>/dev/null; return $rc; }
timeout 70 python bench.py > $O/bench_full.json 2> $O/bench_full.err
python - <<PY
import json
d=json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
print(d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
for k,v in d.get("proof",{}).items(): print(k, {x:v.get(x) for x in ("value","verified_by_oracle","error")})
print(d.get("cpu_baseline"))
PY
echo "bench t=${SECONDS}s"
cd /tmp
run_bounded 30 $O/prof_bench.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $ROOT/bench.py --no-proof --no-cpu-baseline
echo "kernel trace rc=$? t=${SECONDS}s"
run_bounded 25 $O/pmc_fetch.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $ROOT/bench.py --no-proof --no-cpu-baseline --steps 16 --warmup 8
echo "pmc fetch rc=$? t=${SECONDS}s"
run_bounded 25 $O/pmc_write.log rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $ROOT/bench.py --no-proof --no-cpu-baseline --steps 16 --warmup 8
echo "pmc write rc=$? t=${SECONDS}s"
if [ $SECONDS -lt 62 ]; then
    run_bounded 30 $O/prof_sc.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sc -- python $ROOT/bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify
    echo "sc trace rc=$? t=${SECONDS}s"
fi
echo "done t=${SECONDS}s"
