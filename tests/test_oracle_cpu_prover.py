"""CPU: the array-based restated CPU prover (oracle/cpu_prover.py: C primitives + OpenMP, what bench.py times as the
proof-level cpu_baseline) produces the same proof bytes as the big-int restatement the GPU session is byte-equal to."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
S = 0x5EC2E7


@pytest.mark.parametrize("shape,multiopen", [("plain", "gwc"), ("plain", "shplonk"), ("wide", "shplonk"), ("rotations", "gwc"), ("lookups5", "shplonk"), ("two_phase", "shplonk")])
def test_same_bytes_as_the_big_int_prover(shape, multiopen):
    from plonk_fixtures import build_circuit, build_multi_lookup_circuit, build_rotation_circuit
    from oracle import cpu_prover as cp, plonk_prover as pp, plonk_verifier as pv, pairing
    seed = bytes((5 * i + 1) & 0xFF for i in range(16))
    if shape == "two_phase":
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
        import t1_kit
        circ, phase_witness, inst = t1_kit.two_phase_case(6)
        want = pp.create_proof(circ, pp.Srs(circ.k, S), [[0] * circ.n for _ in range(circ.A)], inst, 77, seed, multiopen, phase_witness=phase_witness)
        # the array prover takes all columns up front: replay the phases with the challenges the big-int prover squeezed
        from oracle import transcripts  # noqa: F401
        adv = [None] * circ.A
        box = {}

        def spy(phase, challenges):
            cols = phase_witness(phase, challenges)
            box.update(cols)
            return cols
        pp.create_proof(circ, pp.Srs(circ.k, S), [[0] * circ.n for _ in range(circ.A)], inst, 77, seed, multiopen, phase_witness=spy)
        adv = [box[i] for i in range(circ.A)]
    else:
        circ, adv, inst = {"plain": lambda: build_circuit(6, 1, False), "wide": lambda: build_circuit(7, 2, True),
                           "rotations": lambda: build_rotation_circuit(6, 1), "lookups5": lambda: build_multi_lookup_circuit(7, 1, 5, 1, 9)}[shape]()
        want = pp.create_proof(circ, pp.Srs(circ.k, S), adv, inst, 77, seed, multiopen)
    timings = {}
    key = cp.keygen(circ)
    got = cp.create_proof(circ, cp.Srs(circ.k, S), adv, inst, 77, seed, multiopen, timings=timings, key=key)
    assert got == want
    assert "evaluate_h" in timings and sum(timings.values()) > 0
    # Montgomery arrays as input (how bench.py hands the witness over) give the same bytes
    from oracle import cref
    got2 = cp.create_proof(circ, cp.Srs(circ.k, S), [cref.to_mont([v % cp.R for v in col]) for col in adv], inst, 77, seed, multiopen, key=key)
    assert got2 == want
