#!/bin/bash
# timeline of one steady-state MSM period of the bench command (kernel trace), one-launch scans on and off
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 1 0; do
  rm -rf /tmp/tl$v
  ZK_MSM_SCAN1=$v timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl$v -- python bench.py --no-proof --no-cpu-baseline --steps 32 --warmup 16 > /dev/null 2>&1
  f=$(find /tmp/tl$v -name "*kernel_trace.csv" | head -1)
  echo "== ZK_MSM_SCAN1=$v $f"
  python tools/trace_timeline.py "$f" 60 > gpurun_out/timeline_scan$v.txt 2>&1
  head -70 gpurun_out/timeline_scan$v.txt
done
