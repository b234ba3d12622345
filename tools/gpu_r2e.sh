#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=12 --tb=short -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_full.json 2> $O/bench_full.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_full.json"))
    print(d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", d["extra"]["kernel_avg_ms"])
    for k,v in d.get("proof",{}).items(): print(k, {x:v.get(x) for x in ("value","create_proof_s","verified_by_oracle","error","chain_of_4_proofs_s")}, v.get("roofline_quotient",{}).get("avg_launch_ms"))
except Exception as ex: print("bench failed", ex)
PY
tail -3 $O/bench_full.err
timeout 300 python bench.py --gpus 2 --no-proof --no-cpu-baseline > $O/bench_g2.json 2> $O/bench_g2.err; python -c "
import json; d=json.load(open('$O/bench_g2.json')); print('gpus=2', d['value'], d['n_gpus'], d['config']['parallelism'])"
cd /tmp
ZK_BENCH_HARD_EXIT=1 timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --no-proof --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); echo "stats: $f"; head -24 "$f" | cut -c1-150
