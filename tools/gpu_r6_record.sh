#!/bin/bash
# Round-6 record run (one gpurun call): the driver's bench line; one kernel trace PER PROOF KIND of the EVM-style headline (timed / structure-blind /
# degree-blind / host-memory: --steps 1 --warmup 1 each, side measurements off) and one of the plain shape; the MSM / NTT section; PMC passes (counters only):
# FETCH_SIZE, WRITE_SIZE of the MSM / NTT section and of one headline proof, and the issue counters of one whole proof.
# tools/summarize_r06.py turns gpurun_out/r6rec/ into profiles/r06_*.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r6rec; mkdir -p $O
export TMPDIR=/tmp
# SKIP_BENCH=1: traces and counters only (the bench line of tools/gpu_r6_bench_only.sh stays)
if [ "$SKIP_BENCH" != 1 ]; then rm -rf $O; mkdir -p $O; timeout 1500 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$? t=${SECONDS}"; else rm -rf $O/prof_* $O/pmc_*; fi
cd /tmp
Q="--no-cpu-baseline --no-proof --no-msm-ntt --no-verify"
trace() {  # name, env...
  local name=$1; shift
  env "$@" ZK_BENCH_QUICK=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python $ROOT/bench.py $Q --steps 1 --warmup 1 > $O/prof_$name.log 2>&1
  echo "trace $name rc=$? t=${SECONDS}"
}
trace timed ZK_X=0
trace structure_blind ZK_MSM_RUNS=0 ZK_MSM_DIFF=0
trace degree_blind ZK_QUOTIENT_SPLIT=0 ZK_QUOTIENT_ADDSPLIT=0
trace host ZK_BENCH_KIND=host
trace plain ZK_BENCH_SHAPE=plain
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_msmntt -- python $ROOT/bench.py --only-msm-ntt --no-cpu-baseline > $O/prof_msmntt.log 2>&1
echo "msm/ntt trace rc=$? t=${SECONDS}"
pmc() {  # name, counters, command...
  local name=$1 ctr=$2; shift 2
  ZK_BENCH_QUICK=1 timeout 600 rocprofv3 --pmc $ctr -d $O/pmc_$name --output-format csv -- "$@" > $O/pmc_$name.log 2>&1
  echo "pmc $name rc=$? t=${SECONDS}"
}
pmc fetch FETCH_SIZE python $ROOT/bench.py --only-msm-ntt --no-cpu-baseline
pmc write WRITE_SIZE python $ROOT/bench.py --only-msm-ntt --no-cpu-baseline
pmc fetch_proof FETCH_SIZE python $ROOT/bench.py $Q --steps 1 --warmup 0
pmc write_proof WRITE_SIZE python $ROOT/bench.py $Q --steps 1 --warmup 0
pmc issue_proof "SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" python $ROOT/bench.py $Q --steps 1 --warmup 0
pmc waves_proof "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS" python $ROOT/bench.py $Q --steps 1 --warmup 0
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python - $O <<'PY'
import collections, csv, glob, json, sys
O = sys.argv[1]
for run in ("pmc_fetch", "pmc_write", "pmc_fetch_proof", "pmc_write_proof", "pmc_issue_proof", "pmc_waves_proof"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{O}/{run}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zk::", "")
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    json.dump({f"{k}|{c}": {"sum": v[0], "launches": v[1]} for (k, c), v in acc.items()}, open(f"{O}/{run}_sums.json", "w"), indent=0)
    print(run, len(acc), "kernel/counter pairs")
PY
find $O -name "*counter_collection.csv" -delete
head -c 300 $O/bench_full.json; echo
