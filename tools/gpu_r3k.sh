#!/bin/bash
# round 3, pass k: kernel trace of the SuperCircuit-shape proof (current build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sc -- python $ROOT/bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify > $O/prof_sc.log 2>&1
echo "rc=$?"; du -sh $O; ls $O/prof_sc/*/ | head
cd $O/prof_sc/*/ && gzip -9 *_kernel_trace.csv && ls -la
