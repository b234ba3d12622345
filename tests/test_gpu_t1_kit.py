"""GPU: the T1 kit made THROUGH THE LIBRARY (`tools/t1_kit.py make --gpu`: every self-check proof comes out of libzkmi355.so on the
MI355X) verifies from its files alone and is byte-identical to the kit the oracle's big-int prover makes -- so the vectors a box with
cargo will put in front of upstream `verify_proof` [REF circuit-benchmarks/src/super_circuit.rs:141-154] are the current library's:
three phases with the SuperCircuit's challenges, a three-chunk permutation, two-input lookups, SHPLONK and GWC under Blake2b, SHPLONK
under Poseidon."""
import filecmp
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kit_made_on_the_gpu_equals_the_oracle_made_kit(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda *a: subprocess.run([sys.executable, os.path.join(ROOT, "tools", "t1_kit.py"), *a], env=env, capture_output=True, text=True, timeout=840)
    gpu_kit, cpu_kit = str(tmp_path / "gpu"), str(tmp_path / "cpu")
    res = run("make", gpu_kit, "--gpu")
    assert res.returncode == 0, res.stdout + res.stderr
    res = run("check", gpu_kit)
    assert res.returncode == 0 and res.stdout.count("accepted") == 27 and "REJECTED" not in res.stdout, res.stdout + res.stderr
    res = run("make", cpu_kit)
    assert res.returncode == 0, res.stdout + res.stderr
    cases = sorted(os.listdir(cpu_kit))
    assert "three_phase_k6" in cases and "wide_k7" in cases and "lookup_2x_k6" in cases and sorted(os.listdir(gpu_kit)) == cases
    for case in cases:
        names = sorted(os.listdir(os.path.join(cpu_kit, case)))
        assert {"selfcheck_shplonk.bin", "selfcheck_gwc.bin", "selfcheck_poseidon_shplonk.bin", "desc.txt", "params.bin"} <= set(names)
        for f in names:
            assert filecmp.cmp(os.path.join(cpu_kit, case, f), os.path.join(gpu_kit, case, f), shallow=False), f"{case}/{f} differs between the GPU-made and the oracle-made kit"
