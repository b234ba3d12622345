#!/bin/bash
# PMC passes over tools/quot_evm_loop.py (counters only, one small group per pass): where the evaluator's wave cycles go on the EVM-style program
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-pmc_quot_evm}; mkdir -p $O
for a in 0 1; do timeout 300 python tools/quot_evm_loop.py 20 3 $a 2>&1 | tail -1; done
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU_INT64 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/g$i -o g$i --output-format csv -- python tools/quot_evm_loop.py 20 2 > $O/g$i.log 2>&1
done
python - $O <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/g*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('zk::', '')
        if 'quotient' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
