"""CPU: the committed bench line (`profiles/bench_r06.json`, what `python bench.py` printed on the GPU box) keeps the driver's
contract as the tier framing reads it: a step is one full proof of BASELINE configs[3]'s stand-in, `value` = seconds per proof with
the witness resident in HBM, `roofline` = the dominant kernel class with its algorithmic bytes, `cpu_baseline` = the restated CPU
prover on a BASELINE-size sample; the side records agree with each other."""
import csv
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


@pytest.fixture(scope="module")
def line():
    with open(os.path.join(P, "bench_r06.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_contract_fields(line):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["higher_is_better"] is False and line["unit"] == "s" and line["data"] == "synthetic-shape"
    assert line["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert "workload" in line["config"] and "model" not in line["config"] and "configs[3]" in line["config"]["workload"]
    assert abs(line["value"] * 1e3 - line["ms_per_step"]) < 0.01 * line["ms_per_step"]
    cfg = line["config"]
    assert (cfg["k"], cfg["advice"], cfg["fixed"], cfg["permutation_columns"], cfg["lookups"], cfg["degree"]) == (20, 1000, 150, 150, 100, 9)
    assert cfg["advice_phases"] == 3 and sum(cfg["advice_columns_per_phase"]) == 1000 and cfg["challenges"] == 3
    dist = cfg["witness_cell_distribution"]                 # 60 % zero / 30 % below 2^16 / 10 % uniform, per cell
    assert 0.55 <= dist["zero"] <= 0.70 and 0.20 <= dist["below_2^16"] <= 0.35 and 0.08 <= dist["larger"] <= 0.12
    assert "HBM" in cfg["witness_residency"]
    assert line["extra"]["verified_by_oracle"] is True
    pc = line["extra"]["pcie_inclusive"]                    # the host-memory proof is reported beside `value`, never as it
    assert pc["same_proof_bytes"] is True and pc["value"] > line["value"]
    cb = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("port", "reference") and cb["unit"] == line["unit"] and cb["cores"] >= 1
    assert cb["same_proof_bytes"] is True and "k = 18" in cb["sample"] and cb["gpu_same_sample_s"] < cb["value"]


def test_roofline_is_what_it_says(line):
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # achieved = algorithmic bytes per launch / the measured time per launch (HIP events over the timed region)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    # the class the line calls dominant is the one with the largest device time among those it timed: on the EVM-style shape the
    # evaluator (round 6), whose algorithmic bytes are 32 B x rows x (distinct operands + 1) per class launch
    per_class = line["extra"]["kernel_class_device_ms_per_proof"]
    ntt, quot = per_class["ntt_pass"] + per_class["ntt_last"], per_class.get("quotient_coset", 0)
    assert ("k_quotient_eval" in r["kernel"]) == (quot > ntt) and max(ntt, quot) >= per_class["msm_buckets"]
    assert abs(r["device_ms_per_proof"] - max(ntt, quot)) < 0.02 * max(ntt, quot)
    names = [x["kernel"] for x in line["rooflines"]]
    assert any("k_msm_buckets" in nm for nm in names) and any("k_quotient_eval" in nm for nm in names) and any("k_ntt" in nm for nm in names)
    rn = next(x for x in line["rooflines"] if "k_ntt" in x["kernel"])
    assert rn["algorithmic_bytes_per_launch"] == 64 << 20
    rq = next(x for x in line["rooflines"] if "k_quotient_eval" in x["kernel"])
    ex = rq["executed"]                                        # what the interpreter loads and multiplies: more than the algorithmic bytes, below both roofs
    assert ex["operand_bytes_per_proof"] > rq["algorithmic_bytes_per_proof"] and 0 < ex["frac_of_hbm_peak"] < 1 and 0 < ex["frac_of_product_peak"] < 1
    pr = line["proof_roofline"]
    assert abs(pr["algorithmic_bytes"] - sum(pr["by_stage_bytes"].values())) <= 1
    assert abs(pr["frac"] - pr["algorithmic_bytes"] / line["value"] / 1e9 / 8000.0) < 1e-3 and pr["frac"] < 1.0


def test_the_headline_is_the_adverse_shape(line):
    """round 6: `value` is the EVM-style stand-in (>= 5 000 constraints of degree 5..9 over ~160 step columns, wide lookups); the plain shape
    of rounds 1-5 is reported beside it, measured the same way, and is the faster of the two; both carry `degree_blind`"""
    e = line["extra"]
    assert e["shape"] == "evm" and "EVM-style" in line["config"]["workload"]
    assert sum(v for d_, v in e["gate_degrees"].items() if 5 <= int(d_) <= 9) >= 5000 and e["lookup_tuple_widths"] == [4, 6, 8]
    plan = e["evaluator"]["plan"]
    assert plan["expression_graph"] == 1 and sum(c["instructions"] for c in plan["classes"] if c["used"]) >= 50000
    assert max(c["slots_alive"] for c in plan["classes"]) <= 64
    db = e["degree_blind"]
    assert db["same_proof_bytes"] is True and db["value"] >= line["value"] and db["plan"]["degree_classes"] == 0
    plain = line["proof"]["supercircuit_shape_k20_plain"]
    assert plain["extra"]["shape"] == "plain" and plain["value"] < line["value"] and plain["extra"]["verified_by_oracle"] is True
    assert plain["extra"]["degree_blind"]["same_proof_bytes"] is True and plain["extra"]["degree_blind"]["value"] > plain["value"]
    proj = e["projected_rank_device_s"]
    assert "projection" in proj["note"] or "emulated" in proj["note"]
    assert line["value"] > proj["rank_device_s"]["2"] > proj["rank_device_s"]["4"] > proj["rank_device_s"]["8"] > 0


def test_msm_ntt_section_keeps_configs_1(line):
    m = line["msm_ntt"]
    assert "configs[1]" in m["workload"] and m["msm_mscalar_per_s"] > 100 and m["ntt_gfieldop_per_s"] > 50
    assert abs(m["msm_mscalar_per_s"] - (1 << 20) / (m["ms_per_step"] * 1e-3) / 1e6) < 0.01 * m["msm_mscalar_per_s"]
    rb = m["rooflines"][0]
    assert rb["kernel"] == "k_msm_buckets" and rb["algorithmic_bytes_per_launch"] == 96 << 20
    assert rb["traffic"] is not None and rb["traffic"] > rb["algorithmic_bytes_per_launch"]
    assert m["cpu_baseline"]["unit"] == "Mscalar/s" and m["cpu_baseline"]["kind"] == "port"


def test_every_proof_in_the_line_was_verified(line):
    proofs = line.get("proof") or {}
    assert set(proofs) >= {"keccak_shape_k18", "bundle_shape_k21", "supercircuit_shape_k20_dense", "supercircuit_shape_k20_small", "supercircuit_shape_k20_plain"}
    for name, rec in proofs.items():
        assert not rec.get("error"), name
        if name.endswith("_mock"):
            continue
        assert rec.get("verified_by_oracle", (rec.get("extra") or {}).get("verified_by_oracle")) is True, name
        assert rec["data"] == "synthetic-shape"
    b_ = proofs["bundle_shape_k21"]                         # [REF aggregator/configs/bundle_circuit.config]: degree 21, 5 + 1 advice
    assert (b_["k"], b_["advice"]) == (21, 6) and b_["transcript"] == "poseidon"
    # more field-sized cells cost more: dense > 60/30/10 (single-phase variants aside) > all small
    assert proofs["supercircuit_shape_k20_dense"]["value"] > proofs["supercircuit_shape_k20_small"]["value"]


def test_the_structure_reading_paths_are_reported_beside_the_headline(line):
    """The headline's permutation and lookup commitments read structure out of the witness (csrc/runs.hip): the line says how much
    structure there was, and what the same proof (same bytes) costs with those paths off -- never less than the headline."""
    cfg, blind = line["config"], line["extra"]["structure_blind"]
    assert 0.0 < cfg["rows_with_an_active_lookup"] < 1.0 and 0.0 <= cfg["rows_with_a_copy_constraint_per_column"] < 1.0
    assert blind["same_proof_bytes"] is True and blind["unit"] == "s"
    assert blind["value"] >= line["value"]


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
