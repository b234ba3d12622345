"""CPU: the T1 kit (tools/t1_kit.py + shim/t1_standalone) -- the description file carries the whole circuit, the kit verifies
from its files alone, a second `prove` round under another vk.transcript_repr (what the Rust `repr` step hands back) works,
and the Rust program refers to the files the Python side writes."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_description_file_rebuilds_the_same_key_blob():
    from plonk_fixtures import build_circuit, build_multi_lookup_circuit, build_rotation_circuit
    from zkevm_circuits_amd import plonk
    for circ, _adv, _inst in (build_circuit(6, 1, False), build_circuit(6, 4, True), build_rotation_circuit(6, 1, blinding_factors=15),
                              build_multi_lookup_circuit(6, 1, 1, 2, 3), build_multi_lookup_circuit(6, 1, 2, 1, 3), build_multi_lookup_circuit(6, 2, 5, 1, 9)):
        again = plonk.Circuit.from_kit_desc(circ.kit_desc())
        assert again.blob() == circ.blob()           # same constraint system, queries, permutation columns, sigma and fixed columns, bit for bit
        assert again.kit_desc() == circ.kit_desc()


def test_upstream_derives_the_blinding_factors():
    from plonk_fixtures import build_rotation_circuit
    circ, _, _ = build_rotation_circuit(6, 1)            # built with 17 blinding rows: fine for the GPU suite, not what upstream would derive
    assert circ.halo2_blinding_factors() == 15
    with pytest.raises(AssertionError, match="blinding_factors"):
        circ.kit_desc()


def test_kit_round_trip_from_files_only(tmp_path):
    kit = str(tmp_path / "kit")
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda *a: subprocess.run([sys.executable, os.path.join(ROOT, "tools", "t1_kit.py"), *a], env=env, capture_output=True, text=True, timeout=900)
    res = run("make", kit)
    assert res.returncode == 0, res.stdout + res.stderr
    res = run("check", kit)
    assert res.returncode == 0 and res.stdout.count("accepted") == 27 and "REJECTED" not in res.stdout, res.stdout + res.stderr     # 9 cases x (SHPLONK, GWC, Poseidon + SHPLONK)
    assert run("prove", kit).returncode == 1              # no vk_repr.hex yet: the Rust `repr` step has not run
    # stand in for the Rust step: any field element will do as "upstream's vk.transcript_repr" for the mechanics
    for case in os.listdir(kit):
        open(os.path.join(kit, case, "vk_repr.hex"), "w").write(f"{0x1b3d158be8148c9e8ac9fce6eff2c576027c356ee1ff68ad7662d61556d5a7d7:064x}\n")
    res = run("prove", kit)
    assert res.returncode == 0, res.stdout + res.stderr
    res = run("check", kit)
    assert res.returncode == 0 and res.stdout.count("accepted") == 45, res.stdout + res.stderr
    # a proof made under one repr must not verify under another (the repr is absorbed first)
    d = os.path.join(kit, "plain_k6")
    os.replace(os.path.join(d, "selfcheck_shplonk.bin"), os.path.join(d, "proof_shplonk.bin"))
    res = run("check", kit)
    assert res.returncode == 1 and "plain_k6: proof_shplonk.bin" in res.stdout and "REJECTED" in res.stdout


def test_rust_program_and_python_tool_agree_on_the_files():
    rust = open(os.path.join(ROOT, "shim", "t1_standalone", "src", "main.rs")).read()
    tool = open(os.path.join(ROOT, "tools", "t1_kit.py")).read()
    for f in ("desc.txt", "params.bin", "instances.txt", "vk_repr.hex", "expect_vk_commitments.hex", "proof_shplonk.bin", "proof_gwc.bin"):
        assert f in rust and (f in tool or f.replace("shplonk", "{mo}").replace("gwc", "{mo}") in tool), f
    # every line kind the description can hold is parsed on the Rust side
    from zkevm_circuits_amd import plonk
    kinds = set(re.findall(r'out\.append\(f?"(\w+) ', open(plonk.__file__).read())) | {"k", "fixed", "advice", "instance", "challenges", "degree", "blinding_factors", "end"}
    for kind in kinds:
        assert f'"{kind}" =>' in rust, kind
    assert 'rev = "e5ddf67e5ae16be38d6368ed355c7c41906272ab"' in open(os.path.join(ROOT, "shim", "t1_standalone", "Cargo.toml")).read()


def test_the_kit_covers_what_the_headline_proof_is_made_of():
    """the kit's cases must hold what the reference-held proof cannot pin -- three phases with the SuperCircuit's
    challenges, a multi-chunk permutation, a two-input lookup, GWC, and the Poseidon transcript next to Blake2b."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import t1_kit
    cs = t1_kit.cases()
    three = cs["three_phase_k6"][0]
    assert three.num_phases() == 3 and [three.advice_phase.count(p) for p in range(3)] == [3, 1, 1] and len(three.challenge_phase) == 3
    assert sorted(three.challenge_phase) == [0, 0, 1] and len(three.lookups) == 1 and len(three.lookups[0].inputs[0]) == 2
    wide = cs["wide_k7"][0]
    d = wide.degree()
    assert (len(wide.perm_cols) + d - 3) // (d - 2) >= 3                     # three chunks of the grand product: Z chaining at omega^last
    assert any(len(lk.inputs) == 2 for lk in cs["lookup_2x_k6"][0].lookups)    # two input tuples into one table
    tool = open(os.path.join(ROOT, "tools", "t1_kit.py")).read()
    assert '("shplonk", "gwc")' in tool and "selfcheck_poseidon_shplonk.bin" in tool
