#!/bin/bash
# round 6, second half: stage table of the EVM-style headline proof with the new evaluator; issue counters of the evaluator on the loop tool
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/${1:-r6stages}; mkdir -p $O
export TMPDIR=/tmp
ZK_PROVER_TRACE=1 ZK_BENCH_QUICK=1 timeout 900 python bench.py --no-cpu-baseline --no-proof --no-msm-ntt --no-verify --steps 2 --warmup 1 > $O/evm.json 2> $O/evm.err
python tools/trace_stages.py $O/evm.err 2 2>/dev/null | head -70
cd /tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_INT64 SQ_INSTS_LDS" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/g$i -o g$i --output-format csv -- python $ROOT/tools/quot_evm_loop.py 20 2 > $O/g$i.log 2>&1
done
cd $ROOT
python - $O <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('zk::', '')
        if 'quotient' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:28s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
