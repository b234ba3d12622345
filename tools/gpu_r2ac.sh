#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2ac; mkdir -p $O
export TMPDIR=/tmp
run_bounded() { local secs=$1 log=$2; shift 2; setsid "$@" > "$log" 2>&1 & local pid=$!; ( sleep "$secs"; kill -TERM -- -"$pid" 2>/dev/null; sleep 3; kill -KILL -- -"$pid" 2>/dev/null ) & local wd=$!; wait "$pid"; local rc=$?; kill "$wd" 2>/dev/null; return $rc; }
cd /tmp
run_bounded 200 $O/prof_kc.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kc -- python $ROOT/bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 4 --no-verify
echo rc=$?
python - <<PY
import csv, glob
f=glob.glob("$O/prof_kc/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6, "launches", sum(int(r["Calls"]) for r in rows))
for r in rows[:25]: print(f'{r["Name"][:56]:56s} {int(r["Calls"]):6d} tot {float(r["TotalDurationNs"])/1e6:8.2f} ms avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
