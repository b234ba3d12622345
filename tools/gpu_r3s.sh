#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for lib in "" "$PWD/zkevm-circuits_amd/lib/libzkmi355_r2.so"; do
  echo "lib=${lib:-current}"
  for a in "20 30" "20 16" "20 8" "18 8"; do ZKMI355_LIB=${lib:-$PWD/zkevm-circuits_amd/lib/libzkmi355.so} timeout 120 python tools/msm_narrow.py $a 32 2>&1 | grep "hint=1"; done
done
