#!/bin/bash
# One GPU call: A/B of the arithmetic builds (tools/build_arith_variants.sh), then the full GPU suite, smoke and the
# driver's bench on the build that won.  Everything is bounded: the call has < 4.5 GPU-minutes.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/arith; L=$ROOT/zkevm-circuits_amd/lib; mkdir -p $O
export TMPDIR=/tmp
b() { [ -s $O/bench_$1.json ] || ZKMI355_LIB=$L/libzkmi355_$1.so timeout 60 python bench.py --no-proof --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err; }
q() { ZKMI355_LIB=$L/libzkmi355_$1.so timeout 40 python tools/quot_loop.py 19 100 5 > $O/quot_$1.txt 2>&1; }
suite() { ZKMI355_LIB=$L/libzkmi355_$1.so timeout 120 python -m pytest tests -x -q -m gpu > $O/pytest_$1.log 2>&1; }
val() { python -c "import json,sys; d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1]); e=d['extra']; print('$1', d['value'], 'msm', e['msm_pipelined_ms'], 'lone', e['msm_lone_ms'], 'ntt', e['ntt_only_ms'], e['kernel_avg_ms'])" 2>&1 | tail -1; }
b head; val head
b c000; val c000
b c111; val c111
q c000; q c111; tail -n1 $O/quot_c000.txt $O/quot_c111.txt
CH=$(python tools/pick_arith_variant.py pick $O 2>> $O/pick.err) || CH=c000
echo "picked per group: $CH"
b $CH; val $CH
FINAL=$(python tools/pick_arith_variant.py final $O $CH 2>> $O/pick.err) || FINAL=head
echo "against head: $FINAL  (t=${SECONDS}s)"
if [ "$FINAL" != head ]; then
    suite $FINAL; rc=$?
    echo "suite on $FINAL rc=$rc: $(grep -E 'passed|failed|error' $O/pytest_$FINAL.log | tail -1)  (t=${SECONDS}s)"
    if [ $rc -ne 0 ] && [ "$FINAL" != c000 ] && [ $SECONDS -lt 150 ] && [ "$(python tools/pick_arith_variant.py final $O c000)" = c000 ]; then
        FINAL=c000; suite c000; rc=$?
        echo "suite on c000 rc=$rc: $(grep -E 'passed|failed|error' $O/pytest_c000.log | tail -1)  (t=${SECONDS}s)"
    fi
    [ $rc -ne 0 ] && FINAL=head
fi
echo $FINAL > $O/final.txt
echo "FINAL=$FINAL"
if [ "$FINAL" != head ]; then
    ZKMI355_LIB=$L/libzkmi355_$FINAL.so timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
    if [ $SECONDS -lt 205 ]; then
        ZKMI355_LIB=$L/libzkmi355_$FINAL.so timeout 100 python bench.py > $O/bench_full_$FINAL.json 2> $O/bench_full_$FINAL.err
        python - <<PY
import json
d=json.loads(open("$O/bench_full_$FINAL.json").read().strip().splitlines()[-1])
print(d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
for k,v in d.get("proof",{}).items(): print(k, {x:v.get(x) for x in ("value","create_proof_s","verified_by_oracle","error")}, (v.get("roofline_quotient") or {}).get("frac"))
PY
    fi
fi
echo "done t=${SECONDS}s"
