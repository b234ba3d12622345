"""GPU: `bench.py --gpus 2` end to end, under pytest, so that the first multi-GPU scaling run is not the first execution of the
N > 1 bench path.  The test box has ONE GPU: with the test switch ZK_BENCH_K=12 the two ranks share it (gloo callbacks instead of the
library's RCCL communicator -- RCCL ranks cannot share a device) and run the SAME code as N = 1 over the same column counts: the
three-phase SuperCircuit-shape circuit, witness resident and handed over in place, every rank holding only the columns it owns
(the others arrive device to device), commitments by column, quotient by (degree class, coset).  Required: the line keeps the
driver's contract (steps / warmup as given, n_gpus, roofline from rank 0's events), the proof is accepted by the oracle verifier and
is BYTE-IDENTICAL to the N = 1 proof.  [REF circuit-benchmarks/src/super_circuit.rs:117-132], [REF zkevm-circuits/src/util.rs:120-133]"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(gpus: int, steps: int, warmup: int):
    env = dict(os.environ, ZK_BENCH_K="12", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k_, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", str(steps), "--warmup", str(warmup),
                          "--no-cpu-baseline", "--no-proof", "--no-msm-ntt"], capture_output=True, text=True, timeout=840, env=env, cwd=ROOT)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and lines, res.stdout[-2000:] + res.stderr[-4000:]
    return json.loads(lines[-1])


@pytest.fixture(scope="module")
def single():
    return run_bench(1, 2, 1)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_gpus_n_is_the_same_workload_as_n1(single, world):
    line = run_bench(world, 2, 1)
    assert not line.get("error"), line
    assert (line["n_gpus"], line["steps"], line["warmup"]) == (world, 2, 1) and (single["n_gpus"], single["steps"], single["warmup"]) == (1, 2, 1)
    assert line["metric"] == single["metric"] and line["unit"] == "s" and line["scaling"] == "strong" and line["value"] > 0
    # the same circuit, phases and hand-over: the two lines differ in N only
    for key in ("workload", "k", "advice", "fixed", "permutation_columns", "lookups", "degree", "advice_phases", "advice_columns_per_phase", "challenges", "multiopen", "transcript"):
        assert line["config"][key] == single["config"][key], key
    assert line["config"]["k"] == 12 and line["config"]["advice_phases"] == 3 and "HBM" in line["config"]["witness_residency"]
    assert f"x{world}" in line["config"]["parallelism"]
    # the roofline record is there at every N (rank 0's events), the CPU baseline at N = 1 only
    for rec in (line, single):
        r = rec["roofline"]
        assert r and r["bound"] == "hbm" and "k_ntt" in r["kernel"] and r["transforms_per_proof"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert "cpu_baseline" in line and line["cpu_baseline"] is None
    # one proof, however many ranks made it
    assert single["extra"]["verified_by_oracle"] is True and line["extra"]["verified_by_oracle"] is True
    assert line["extra"]["proof_sha256"] == single["extra"]["proof_sha256"] and line["extra"]["proof_bytes"] == single["extra"]["proof_bytes"]
    assert line["extra"]["structure_blind"]["same_proof_bytes"] is True
    # a sharded rank transforms fewer (column, coset) pairs than the single GPU does
    assert line["roofline"]["transforms_per_proof"] < single["roofline"]["transforms_per_proof"]
