#!/bin/bash
# round 5: where the device idles inside a headline proof (kernel trace -> tools/kernel_gaps.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/${1:-r5gaps}; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
ZK_BENCH_NO_EXTRAS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $ROOT/bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify --steps 2 --warmup 1 > $O/bench.log 2>&1
echo "rc=$?"; tail -c 300 $O/bench.log; echo
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/kernel_gaps.py "$f" | tee $O/gaps.txt
s=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$s" $O/kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete
