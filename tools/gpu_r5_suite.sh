#!/bin/bash
# round 5: the -m gpu suite without -x (every failure listed, slowest tests named) + smoke
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/${1:-r5suite}; mkdir -p $O
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -q -m gpu --durations=12 ${@:2} > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|error" $O/pytest.log | tail -5
grep -n "^FAILED\|^ERROR" $O/pytest.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
