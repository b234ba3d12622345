#!/bin/bash
# Round-3 record run of the shipped build on ONE box: the driver's bench line; rocprofv3 kernel statistics and PMC traffic
# (FETCH_SIZE / WRITE_SIZE, separate passes, no trace domains next to --pmc) of the same command; kernel statistics of the
# SuperCircuit-shape proof; PMC traffic of the quotient evaluator (tools/quot_loop.py and the proof itself); NTT issue counters.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3rec; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
run_bounded() { local secs=$1 log=$2; shift 2; setsid "$@" > "$log" 2>&1 & local pid=$!; ( sleep "$secs"; kill -TERM -- -"$pid" 2>/dev/null; sleep 3; kill -KILL -- -"$pid" 2>/dev/null ) & local wd=$!; wait "$pid"; local rc=$?; kill "$wd" 2>/dev/null; return $rc; }
timeout 240 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
python - <<PY
import json
d=json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
print(d["value"], "Mscalar/s", d["ms_per_step"], "ms/step", d["extra"]["kernel_avg_ms"], "lone", d["extra"]["msm_lone_ms"])
for k,v in d.get("proof",{}).items(): print(k, {x:v.get(x) for x in ("value","verified_by_oracle","error","gpu_s","same_proof_bytes")}, (v.get("cpu_baseline") or {}).get("value"))
print(d.get("cpu_baseline"))
PY
echo "bench t=${SECONDS}s"
cd /tmp
run_bounded 60 $O/prof_bench.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $ROOT/bench.py --no-proof --no-cpu-baseline --steps 32 --warmup 16
echo "kernel trace rc=$? t=${SECONDS}s"
run_bounded 60 $O/pmc_fetch.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $ROOT/bench.py --no-proof --no-cpu-baseline --steps 16 --warmup 8
echo "pmc fetch rc=$? t=${SECONDS}s"
run_bounded 60 $O/pmc_write.log rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $ROOT/bench.py --no-proof --no-cpu-baseline --steps 16 --warmup 8
echo "pmc write rc=$? t=${SECONDS}s"
run_bounded 90 $O/prof_sc.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sc -- python $ROOT/bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 --no-verify
echo "sc trace rc=$? t=${SECONDS}s"
for c in FETCH_SIZE WRITE_SIZE; do
  run_bounded 60 $O/pmc_quot_$c.log rocprofv3 --pmc $c --kernel-include-regex "k_quotient_eval" --output-format csv -d $O/pmc_quot_$c -- python $ROOT/tools/quot_loop.py 20 100 3
  echo "quot_loop $c rc=$? t=${SECONDS}s"
  run_bounded 120 $O/pmc_scq_$c.log rocprofv3 --pmc $c --kernel-include-regex "k_quotient_eval" --output-format csv -d $O/pmc_scq_$c -- python $ROOT/bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 1 --no-verify
  echo "sc quotient $c rc=$? t=${SECONDS}s"
done
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"; do
  i=$((i+1))
  run_bounded 40 $O/pmc_ntt_g$i.log rocprofv3 --pmc $grp --kernel-include-regex "k_ntt" --output-format csv -d $O/pmc_ntt_g$i -- python $ROOT/tools/ntt_loop.py 20 6
done
echo "ntt counters t=${SECONDS}s"
cd $ROOT; timeout 60 python tools/ntt_sizes.py > $O/ntt_sizes.txt 2>&1; tail -n 6 $O/ntt_sizes.txt
# keep what travels back small: the kernel traces of the two --kernel-trace passes are large
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O; echo "done t=${SECONDS}s"
