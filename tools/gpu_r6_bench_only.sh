#!/bin/bash
# the driver's bench line alone (the record run's first step), into gpurun_out/r6rec/bench_full.json
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r6rec; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$? t=${SECONDS}"
head -c 400 $O/bench_full.json; echo
