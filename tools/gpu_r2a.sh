#!/bin/bash
# round-2 GPU pass A: parity tests, the driver's bench line, window-size sweep of the merged MSM
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --tb=short -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for c in 16 18 20; do
  ZK_MSM_C=$c timeout 300 python bench.py --no-proof --no-cpu-baseline --steps 32 --warmup 16 > $O/bench_c$c.json 2> $O/bench_c$c.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_c$c.json")); e=d["extra"]
    print("c=$c", d["value"], "Mscalar/s", d["ms_per_step"], "ms/step | sort/buckets/combine/reduce/ntt:", e["kernel_avg_ms"].get("msm_sort"), e["kernel_avg_ms"].get("msm_buckets"), e["kernel_avg_ms"].get("msm_combine"), e["kernel_avg_ms"].get("msm_reduce"), e["ntt_only_ms"], "windows", d["config"]["msm_windows"])
except Exception as ex:
    print("c=$c failed", ex)
PY
done
( time timeout 900 python bench.py ) > $O/bench_full.json 2> $O/bench_full.err
tail -c 3000 $O/bench_full.json
timeout 300 python bench.py --gpus 2 --no-proof --no-cpu-baseline > $O/bench_g2.json 2> $O/bench_g2.err; tail -c 600 $O/bench_g2.json; tail -3 $O/bench_g2.err
