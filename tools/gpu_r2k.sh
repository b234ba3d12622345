#!/bin/bash
# pass K: where the expression evaluator's time goes -- A/B against the previous build, PMC counters
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2k; mkdir -p $O
export TMPDIR=/tmp
PREV=$ROOT/zkevm-circuits_amd/lib_prev/libzkmi355.so
ZKMI355_LIB=$PREV timeout 200 python tools/quot_loop.py 20 100 4 2>&1 | tail -1 | sed 's/^/prev: /'
ZK_QUOTIENT_FUSE=1 timeout 200 python tools/quot_loop.py 20 100 4 2>&1 | tail -1
ZK_QUOTIENT_FUSE=0 timeout 200 python tools/quot_loop.py 20 100 4 2>&1 | tail -1
ZK_QUOTIENT_FUSE=1 timeout 200 python tools/quot_loop.py 18 100 4 2>&1 | tail -1
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAVES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp -d $O/pmc_new/g$i -o g$i --output-format csv -- python $ROOT/tools/quot_loop.py 20 100 2 > $O/pmc_new_g$i.log 2>&1
  ZKMI355_LIB=$PREV timeout 200 rocprofv3 --pmc $grp -d $O/pmc_prev/g$i -o g$i --output-format csv -- python $ROOT/tools/quot_loop.py 20 100 2 > $O/pmc_prev_g$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for tag in ("new", "prev"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/pmc_%s/g*/*counter_collection.csv" % tag):
        for r in csv.DictReader(open(f)):
            if 'k_quotient_eval' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(tag)
    for c, v in sorted(acc.items()):
        print(f"   {c:26s} {sum(v)/len(v):16.0f}  per wave {sum(v)/len(v)/16384:10.1f} (n={len(v)})")
PY
