#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
for bits in 30 0; do
  ZK_BENCH_SCALAR_BITS=$bits timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench_b$bits.json 2> $O/bench_b$bits.err
  python - <<PY
import json
d=json.load(open("$O/bench_b$bits.json")); e=d["extra"]
print("bits=$bits", d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", e["kernel_avg_ms"])
PY
done
cd /tmp
ZK_BENCH_SCALAR_BITS=30 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_sparse -- python $GRAFT_REPO_ROOT/bench.py --no-proof --no-cpu-baseline --steps 32 --warmup 16 > $GRAFT_REPO_ROOT/$O/prof_sparse.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_sparse -name "*kernel_stats.csv" | head -1); echo $f; head -30 $f | cut -c1-160
