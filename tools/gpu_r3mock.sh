#!/bin/bash
# kernel statistics of zk_mock_verify on the SuperCircuit shape (three clean runs + one with a changed cell) and on the k = 14 stand-in
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3mock; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sc -- python $ROOT/bench.py --proof-worker supercircuit_shape_k20_mock > $O/sc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k14 -- python $ROOT/bench.py --proof-worker evm_shape_k14_mock > $O/k14.log 2>&1
for d in sc k14; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp "$f" $O/${d}_kernel_stats.csv
  find $O/$d -name "*kernel_trace.csv" -delete; find $O/$d -name "*.db" -delete
  echo "== $d"; tail -1 $O/$d.log | cut -c1-400; head -14 $O/${d}_kernel_stats.csv
done
