"""CPU: the committed bench line (`profiles/bench_r02_final.json`, what `python bench.py` printed on the GPU box) keeps the
driver's contract, and its numbers agree with the rocprofv3 summaries committed next to it: the dominant kernel's average
launch time inside bench.py (HIP events) against `rocprofv3 --kernel-trace --stats` of the same command, the algorithmic
bytes behind `roofline.achieved`, the PMC traffic behind `roofline.traffic`."""
import csv
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


@pytest.fixture(scope="module")
def line():
    with open(os.path.join(P, "bench_r02_final.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_contract_fields(line):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak" and line["data"] == "synthetic"
    assert line["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert "workload" in line["config"] and "model" not in line["config"]
    # value = scalars per second over the timed steps: 2^20 scalars per step
    assert abs(line["value"] - (1 << 20) / (line["ms_per_step"] * 1e-3) / 1e6) < 0.01 * line["value"]
    cb = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("port", "reference") and cb["unit"] == line["unit"] and cb["cores"] >= 1


def test_roofline_is_what_it_says(line):
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # achieved = algorithmic bytes per launch (96 B x 2^20) / the kernel's average launch time
    assert r["algorithmic_bytes_per_launch"] == 96 << 20
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    # traffic = FETCH_SIZE + WRITE_SIZE per launch from the committed PMC passes
    with open(os.path.join(P, "traffic_r02.json")) as f:
        traffic = json.load(f)
    assert r["traffic"] is not None and abs(r["traffic"] - traffic["msm_buckets_bytes_per_launch"]) <= 0.05 * r["traffic"]
    assert r["traffic"] > r["algorithmic_bytes_per_launch"]


def test_kernel_time_agrees_with_the_rocprof_summary(line):
    rows = list(csv.DictReader(open(os.path.join(P, "r02_prof_bench_kernel_stats.csv"))))
    bk = [r_ for r_ in rows if "k_msm_buckets" in r_["Name"]]
    assert len(bk) == 1
    rocprof_ms = float(bk[0]["AverageNs"]) / 1e6
    assert abs(rocprof_ms - line["roofline"]["avg_launch_ms"]) < 0.05 * rocprof_ms, (rocprof_ms, line["roofline"]["avg_launch_ms"])


def test_every_proof_in_the_line_was_verified(line):
    proofs = line.get("proof") or {}
    assert set(proofs) >= {"keccak_shape_k18", "recursion_shape_k22", "supercircuit_shape_k20"}
    for name, rec in proofs.items():
        assert rec.get("verified_by_oracle") is True and not rec.get("error"), name
        assert rec["data"] == "synthetic-shape"


def test_gpus_flag_without_a_launcher_starts_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset re-executes itself under torch.distributed.run on 127.0.0.1 with N ranks and
    hands its own flags through; under a launcher (WORLD_SIZE set) it must not."""
    import argparse
    import importlib
    import sys

    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd):
        seen["cmd"] = cmd
        return 0

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    args = argparse.Namespace(gpus=4, steps=7, warmup=3, batch=16, no_cpu_baseline=True, no_proof=False)
    assert bench.relaunch_under_launcher(args) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail[:6] == ["--gpus", "4", "--steps", "7", "--warmup", "3"] and "--no-cpu-baseline" in tail and "--no-proof" not in tail
