#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3r; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --proof-worker keccak_shape_k16_cpu_vs_gpu > $O/w.json 2> $O/w.err; echo "rc=$?"; tail -3 $O/w.err; cat $O/w.json | tail -1
