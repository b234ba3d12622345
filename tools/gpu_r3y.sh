#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_msm.py -x -q -m gpu 2>&1 | tail -2
run() { echo "$*"; env "$@" timeout 300 python bench.py --no-proof --no-cpu-baseline --steps 64 --warmup 32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
B=$PWD/zkevm-circuits_amd/lib
run ZKMI355_LIB=$B/libzkmi355_base.so
run ZKMI355_LIB=$B/libzkmi355_prio.so
run ZKMI355_LIB=$B/libzkmi355_base.so
run ZKMI355_LIB=$B/libzkmi355_prio.so
rm -rf /tmp/tl1
ZKMI355_LIB=$B/libzkmi355_prio.so timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl1 -- python bench.py --no-proof --no-cpu-baseline --steps 32 --warmup 16 > /dev/null 2>&1
f=$(find /tmp/tl1 -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py "$f" 60 > gpurun_out/timeline_prio.txt 2>&1
head -40 gpurun_out/timeline_prio.txt
