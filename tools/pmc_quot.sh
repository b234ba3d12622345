#!/bin/bash
# PMC passes over tools/quot_loop.py (counters only, one small group per pass): where the evaluator's wave cycles go
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU_INT64 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d gpurun_out/pmc_quot/g$i -o g$i --output-format csv -- python tools/quot_loop.py 20 100 3 > gpurun_out/pmc_quot_g$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_quot/g*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('zk::', '')
        if 'quotient' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
timeout 100 python tools/quot_loop.py 20 100 5 2>&1 | tail -1
