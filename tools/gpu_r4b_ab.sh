#!/bin/bash
# Round-4 (second session) A/B pass: correctness of the changed kernels, then the library as it was at the start of the session
# (lib/libzkmi355_base.so, built from f6077c2) against the new one on the MSM distribution probe, the evaluator loop and the
# headline proof.  Writes gpurun_out/r4b/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r4b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
BASE=$ROOT/zkevm-circuits_amd/lib/libzkmi355_base.so
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_quotient.py tests/test_gpu_proof.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$? t=${SECONDS}"; tail -3 $O/pytest.log
for v in new base; do
  [ $v = base ] && export ZKMI355_LIB=$BASE || unset ZKMI355_LIB
  timeout 200 python tools/msm_dist.py 20 32 survey_60_30_10,one_percent,small16 > $O/msm_dist_$v.log 2>&1; echo "msm_dist $v rc=$? t=${SECONDS}"
  timeout 120 python tools/quot_loop.py 20 100 5 > $O/quot_$v.log 2>&1; echo "quot $v rc=$? t=${SECONDS}"
done
unset ZKMI355_LIB
ZK_MSM_GM_SLAB=0 timeout 200 python tools/msm_dist.py 20 32 survey_60_30_10 > $O/msm_dist_noslab.log 2>&1; echo "noslab rc=$? t=${SECONDS}"
ZK_MSM_GM_REDG=8 timeout 200 python tools/msm_dist.py 20 32 survey_60_30_10 > $O/msm_dist_redg8.log 2>&1; echo "redg8 rc=$? t=${SECONDS}"
for v in new base; do
  [ $v = base ] && export ZKMI355_LIB=$BASE || unset ZKMI355_LIB
  ZK_PROVER_TRACE=1 timeout 400 python bench.py --no-cpu-baseline --no-proof --no-msm-ntt > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$? t=${SECONDS}"
  head -c 300 $O/bench_$v.json; echo
done
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_msm -- python $ROOT/tools/msm_dist.py 20 32 survey_60_30_10 > $O/prof_msm.log 2>&1; echo "prof rc=$? t=${SECONDS}"
cd $ROOT
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
f=$(find $O/prof_msm -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-160
tail -n 12 $O/msm_dist_new.log $O/msm_dist_base.log $O/msm_dist_noslab.log $O/msm_dist_redg8.log $O/quot_new.log $O/quot_base.log
