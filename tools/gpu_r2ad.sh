#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); O=$ROOT/gpurun_out/r2ad; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_proof.py tests/test_gpu_sharded_ntt.py tests/test_gpu_sharded_proof.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|Error" $O/pytest.log | tail -3
ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --keccak --k 18 --shplonk --pinned --repeat 4 --no-verify > $O/kc.json 2> $O/kc.log
python -c "
import json; d=json.load(open('$O/kc.json')); print('keccak shape', d['create_proof_s'])"
grep "zk prover" $O/kc.log | grep -v "quotient: program" | tail -24 | grep "cosets\|coefficient\|advice\|ifft" | tail -12
ZK_PROVER_TRACE=1 timeout 400 python bench_proof.py --k 20 --shape 1000,150,150,100,9 --shplonk --pinned --repeat 3 --no-verify > $O/sc.json 2> $O/sc.log
python -c "
import json; d=json.load(open('$O/sc.json')); print('sc shape', d['create_proof_s'])"
grep "zk prover" $O/sc.log | grep "cosets\|coefficient" | tail -9
timeout 300 python bench.py --no-proof --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['extra']['kernel_avg_ms'])"
