#!/bin/bash
# last record of the round: the driver's bench line and the kernel statistics of the same command from one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2last; mkdir -p $O
export TMPDIR=/tmp
timeout 38 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$? t=${SECONDS}s"
tail -c 600 $O/bench_full.json | head -c 300; echo
cd /tmp
timeout 12 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $ROOT/bench.py --no-proof --no-cpu-baseline > $O/prof_bench.log 2>&1; echo "trace rc=$? t=${SECONDS}s"
