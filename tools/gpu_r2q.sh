#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2q; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/with_torch.py <<'PY'
import sys, runpy
import torch
torch.cuda.init(); torch.cuda.synchronize()
x = torch.zeros(16, device="cuda")
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
PY
ZK_PROVER_TRACE=1 timeout 400 python /tmp/with_torch.py bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 2 --no-verify > $O/sc_torch.json 2> $O/sc_torch.log
grep "advice upload" $O/sc_torch.log | tail -2
python -c "
import json; d=json.load(open('$O/sc_torch.json')); print('torch loaded first', d['create_proof_s'])"
timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'])"
