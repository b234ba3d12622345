#!/bin/bash
# round 3, pass f: kernel trace (with timestamps) of the bench command on the current build
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $ROOT/bench.py --no-proof --no-cpu-baseline --steps 32 --warmup 16 > $O/prof_bench.log 2>&1
echo "rc=$?"; tail -n 2 $O/prof_bench.log | cut -c1-300
find $O/prof_bench -name "*.csv" | head; du -sh $O
