#!/bin/bash
# round 3, pass a: GPU suite on the tree + smoke + baseline bench line of this box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --steps 32 --warmup 16 --no-proof > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json
