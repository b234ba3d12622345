#!/bin/bash
# kernel statistics of one command: tools/gpu_prof_cmd.sh <tag> <command...>  ->  gpurun_out/<tag>_kernel_stats.csv (top rows printed)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); TAG=$1; shift
O=$ROOT/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
PYTHONPATH=$ROOT timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- "$@" > $O/cmd.log 2>&1
echo "rc=$?"
f=$(find $O -name "*kernel_stats.csv" | head -1)
cp "$f" $ROOT/gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'].split('(')[0].replace('void ','').replace('zk::','')[:60]:60s} calls {int(r['Calls']):6d}  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
PY
tail -3 $O/cmd.log
