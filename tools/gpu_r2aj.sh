#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2aj; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_api_edges.py tests/test_gpu_comm.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|Error" $O/pytest.log | tail -3
timeout 600 python bench.py --gpus 2 --steps 8 --warmup 4 > $O/bench2.json 2> $O/bench2.err; echo "bench --gpus 2 rc=$?"; tail -c 600 $O/bench2.json; tail -3 $O/bench2.err
