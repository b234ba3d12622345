#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_ntt
timeout 600 bash tools/pmc_ntt.sh > gpurun_out/pmc_ntt_r02.txt 2>&1; tail -50 gpurun_out/pmc_ntt_r02.txt
