#!/bin/bash
# profiles of round 2: kernel trace of the bench line, PMC traffic of the MSM / NTT kernels, kernel trace of the SuperCircuit-shape proof
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
run_bounded() {   # run_bounded <seconds> <logfile> cmd...: own process group, killed as a group at the deadline
  local secs=$1 log=$2; shift 2
  setsid "$@" > "$log" 2>&1 &
  local pid=$!
  ( sleep "$secs"; kill -TERM -- -"$pid" 2>/dev/null; sleep 3; kill -KILL -- -"$pid" 2>/dev/null ) &
  local wd=$!
  wait "$pid"; local rc=$?
  kill "$wd" 2>/dev/null
  return $rc
}
cd /tmp
run_bounded 150 $O/prof_bench.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $ROOT/bench.py --no-proof --no-cpu-baseline
echo "kernel trace rc=$?"
run_bounded 150 $O/pmc_fetch.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $ROOT/bench.py --no-proof --no-cpu-baseline --steps 16 --warmup 8
echo "pmc fetch rc=$?"
run_bounded 150 $O/pmc_write.log rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $ROOT/bench.py --no-proof --no-cpu-baseline --steps 16 --warmup 8
echo "pmc write rc=$?"
run_bounded 200 $O/prof_sc.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sc -- python $ROOT/bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify
echo "sc trace rc=$?"
cd $ROOT
find $O -name "*.csv" | head -20
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-140
