#!/bin/bash
# round 6: evaluator kernel with the accumulator in memory / operands unpacked on arrival -- parity, loop timings, instruction counters
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/${1:-r6quot4}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_quotient.py tests/test_gpu_evm_shape.py tests/test_gpu_proof.py tests/test_gpu_mock.py -q -m gpu -x -k "not k20" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
L=$(pwd)/zkevm-circuits_amd/lib
for v in "" _base; do echo "== lib$v"; ZKMI355_LIB=$L/libzkmi355$v.so timeout 300 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1; done
for a in 0 1; do echo "== ACC_MEM=$a"; ZK_QUOTIENT_ACC_MEM=$a timeout 300 python tools/quot_evm_loop.py 20 4 2>&1 | tail -1; done
echo "== 1 buffer"; timeout 300 python tools/quot_evm_loop.py 20 4 1 2>&1 | tail -1
for t in 20; do timeout 200 python tools/quot_loop.py 20 100 5 2>&1 | tail -1; ZKMI355_LIB=$L/libzkmi355_base.so timeout 200 python tools/quot_loop.py 20 100 5 2>&1 | tail -1; done
cd /tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_INT64 SQ_INSTS_LDS -d $O/pmc -o g --output-format csv -- python tools/quot_evm_loop.py 20 2 > $O/pmc.log 2>&1
python - $O <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('zk::', '')
        if 'quotient' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:28s} {sum(v)/len(v):18.0f}  per wave {sum(v)/len(v)/16384:12.0f} (n={len(v)})")
PY
