#!/bin/bash
# kernel statistics of the FINAL round-3 tree: the bench command and the SuperCircuit-shape proof (the PMC passes of
# tools/gpu_r3_record.sh were taken on an earlier build of the same kernels and are not repeated)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3fin; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $ROOT/bench.py --no-proof --no-cpu-baseline --steps 32 --warmup 16 > $O/prof_bench.log 2>&1
echo "bench trace rc=$? t=${SECONDS}"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sc -- python $ROOT/bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 --no-verify > $O/prof_sc.log 2>&1
echo "sc trace rc=$? t=${SECONDS}"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
tail -1 $O/prof_sc.log | cut -c1-300
