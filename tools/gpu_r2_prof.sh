#!/bin/bash
# End-of-round profile pass on the shipped build: kernel statistics and PMC traffic of the driver's bench command, kernel statistics
# of the SuperCircuit-shape proof, batched NTT sizes.  Bounded to ~2 GPU-minutes.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2zz; mkdir -p $O
export TMPDIR=/tmp
run_bounded() { local secs=$1 log=$2; shift 2; setsid "$@" > "$log" 2>&1 & local pid=$!; ( sleep "$secs"; kill -TERM -- -"$pid" 2>/dev/null; sleep 3; kill -KILL -- -"$pid" 2>/dev/null ) & local wd=$!; wait "$pid"; local rc=$?; kill "$wd" 2>/dev/null; return $rc; }
cd /tmp
run_bounded 45 $O/prof_bench.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $ROOT/bench.py --no-proof --no-cpu-baseline
echo "kernel trace rc=$? t=${SECONDS}s"; tail -n 3 $O/prof_bench.log | cut -c1-400
run_bounded 40 $O/pmc_fetch.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $ROOT/bench.py --no-proof --no-cpu-baseline --steps 16 --warmup 8
echo "pmc fetch rc=$? t=${SECONDS}s"
run_bounded 40 $O/pmc_write.log rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $ROOT/bench.py --no-proof --no-cpu-baseline --steps 16 --warmup 8
echo "pmc write rc=$? t=${SECONDS}s"
if [ $SECONDS -lt 75 ]; then
    run_bounded 55 $O/prof_sc.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sc -- python $ROOT/bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify
    echo "sc trace rc=$? t=${SECONDS}s"
fi
cd $ROOT
if [ $SECONDS -lt 115 ]; then timeout 20 python tools/ntt_sizes.py > $O/ntt_sizes.txt 2>&1; tail -n 8 $O/ntt_sizes.txt; fi
echo "done t=${SECONDS}s"
