#!/bin/bash
# Round-5 record run (one gpurun call): the driver's bench line, kernel statistics of the headline proof and of the MSM / NTT
# section, PMC traffic (FETCH_SIZE, WRITE_SIZE: separate passes, counters only) of the MSM / NTT section and of one headline proof.
# tools/summarize_r05.py turns gpurun_out/r5rec/ into profiles/r05_*.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r5rec; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$? t=${SECONDS}"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_proof -- python $ROOT/bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify > $O/prof_proof.log 2>&1
echo "proof trace rc=$? t=${SECONDS}"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_msmntt -- python $ROOT/bench.py --only-msm-ntt --no-cpu-baseline > $O/prof_msmntt.log 2>&1
echo "msm/ntt trace rc=$? t=${SECONDS}"
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python $ROOT/bench.py --only-msm-ntt --no-cpu-baseline > $O/pmc_fetch.log 2>&1
echo "pmc fetch rc=$? t=${SECONDS}"
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python $ROOT/bench.py --only-msm-ntt --no-cpu-baseline > $O/pmc_write.log 2>&1
echo "pmc write rc=$? t=${SECONDS}"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_proof --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify --steps 1 --warmup 0 > $O/pmc_fetch_proof.log 2>&1
echo "pmc fetch proof rc=$? t=${SECONDS}"
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_proof --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify --steps 1 --warmup 0 > $O/pmc_write_proof.log 2>&1
echo "pmc write proof rc=$? t=${SECONDS}"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
# the per-dispatch counter files of the proof passes are tens of MiB: keep per-kernel sums only
python - $O <<'PY'
import collections, csv, glob, json, sys
O = sys.argv[1]
for run in ("pmc_fetch", "pmc_write", "pmc_fetch_proof", "pmc_write_proof"):
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
head -c 400 $O/bench_full.json; echo
