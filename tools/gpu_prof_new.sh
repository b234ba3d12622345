#!/bin/bash
# kernel statistics of one command under the CURRENT library: tools/gpu_prof_new.sh <tag> <command...>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); tag=$1; shift; O=$ROOT/gpurun_out/$tag; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
unset ZKMI355_LIB
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- "$@" > $O/cmd.log 2>&1; echo "prof rc=$? t=${SECONDS}"
cd $ROOT
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
tail -5 $O/cmd.log
python - $O <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/prof/**/*kernel_stats.csv',recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i>=40: break
    n=r['Name'].split('(')[0].replace('void ','').replace('zk::','')
    print(f"{n[:45]:45s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:9.2f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
