"""CPU: the oracle's restatement of halo2's dev::MockProver::verify_at_rows_par (oracle/plonk_verifier.py:mock_failures) on the
fixture circuits, and the challenges MockProver hands a circuit through the C ABI against the constant the reference pins
[REF zkevm-circuits/src/super_circuit.rs:729] (golden G3)."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import bn254 as b  # noqa: E402
from oracle import plonk_verifier as pv  # noqa: E402
from plonk_fixtures import build_circuit, build_multi_lookup_circuit, build_rotation_circuit  # noqa: E402
from zkevm_circuits_amd import plonk  # noqa: E402

R = plonk.R_MOD


def test_satisfied_witnesses_have_no_failures():
    for circ, adv, inst in (build_circuit(6, 1, False), build_circuit(7, 2, True), build_rotation_circuit(6, 1), build_multi_lookup_circuit(6, 1, 2, 2, 3)):
        assert pv.check_witness(circ, adv, inst) is None
        assert pv.mock_failures(circ, adv, inst) == []


def test_a_broken_gate_a_broken_lookup_and_a_broken_copy_are_located():
    circ, adv, inst = build_circuit(6, 1, False)
    rows = [r for r in range(circ.u) if circ.fixed[0][r] == 1]                      # q_mul rows: c = a * b
    adv = [list(c) for c in adv]
    r0 = rows[0]
    adv[2][r0] = (adv[2][r0] + 1) % R
    got = pv.mock_failures(circ, adv, inst)
    assert (pv.MOCK_GATE, 0, 0, r0) in got
    # every record names a gate that reads the changed cell (c, c.next, c.prev) or a copy of it
    for kind, index, sub, row in got:
        assert kind in (pv.MOCK_GATE, pv.MOCK_PERMUTATION)
        if kind == pv.MOCK_GATE:
            assert (index, row) in ((0, r0), (1, r0 - 1), (2, r0 + 1))
    # verify_at_rows: only the listed gate rows are looked at
    assert [f for f in pv.mock_failures(circ, adv, inst, gate_rows=[r0 + 3, r0 + 4]) if f[0] == pv.MOCK_GATE] == []
    # a lookup input that is in no table row
    circ, adv, inst = build_circuit(6, 1, False)
    adv = [list(c) for c in adv]
    lrow = next(r for r in range(circ.u) if circ.fixed[3][r] == 1)
    adv[1][lrow] = (adv[1][lrow] + 5) % R
    got = pv.mock_failures(circ, adv, inst)
    assert got == [(pv.MOCK_LOOKUP, 0, 0, lrow)]
    assert pv.mock_failures(circ, adv, inst, lookup_rows=[r for r in range(circ.u) if r != lrow]) == []
    # a public input that differs from the cell it is copied from: both ends of the cycle are reported
    circ, adv, inst = build_circuit(6, 1, False)
    inst = [list(c) for c in inst]
    inst[0][0] = (inst[0][0] + 1) % R
    got = pv.mock_failures(circ, adv, inst)
    assert len(got) == 2 and all(f[0] == pv.MOCK_PERMUTATION for f in got)
    cols = {circ.perm_cols[f[1]] for f in got}
    assert (plonk.INSTANCE, 0) in cols and (plonk.ADVICE, 1) in cols


def test_mapping_and_sigma_columns_agree():
    circ, adv, inst = build_circuit(7, 2, True)
    mapping = circ.permutation_mapping()
    sig = circ.sigma_columns()
    w = circ.omega()
    rng = random.Random(1)
    for _ in range(50):
        j, i = rng.randrange(len(circ.perm_cols)), rng.randrange(circ.n)
        j2, i2 = mapping[j][i]
        assert sig[j][i] == pow(plonk.FR_DELTA, j2, R) * pow(w, i2, R) % R


def test_mock_challenges_through_the_abi_match_the_pinned_constant():
    from zkevm_circuits_amd import binding
    ch = binding.mock_challenges(5)
    import numpy as np
    vals = [int.from_bytes(np.ascontiguousarray(c).tobytes(), "little") * pow(1 << 256, -1, R) % R for c in ch]
    assert vals[2] == 0x207A52BA34E1ED068BE1E33B0BC39C8EDE030835F549FE5C0DBE91DCE97D17D2          # [REF zkevm-circuits/src/super_circuit.rs:729]
    assert vals == [b.mock_prover_challenge(i + 1) for i in range(5)]
