#!/bin/bash
# what the driver runs at round end: smoke, then the bench line with its own step counts (wall-clock of the whole command printed)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/driver_cmd; mkdir -p $O
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall=${SECONDS}s"
head -c 600 $O/bench.json; echo
