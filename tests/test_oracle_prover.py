"""CPU: the oracle's own prover (oracle/plonk_prover.py, a big-int restatement of halo2's
create_proof) against the oracle's pairing verifier -- the whole path without a GPU.  The GPU
session is compared byte for byte against this prover in tests/test_gpu_proof.py."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import pairing as pr  # noqa: E402
from oracle import plonk_prover as pp  # noqa: E402
from oracle import plonk_verifier as pv  # noqa: E402
from plonk_fixtures import build_circuit, build_rotation_circuit  # noqa: E402

S_SECRET = 0x5EC2E7


@pytest.mark.parametrize("multiopen", ["gwc", "shplonk"])
def test_oracle_prover_is_accepted_by_oracle_verifier(multiopen):
    circ, adv, inst = build_circuit(5, seed=2, wide=True)
    srs = pp.Srs(circ.k, S_SECRET)
    vk_points, vk_repr = pp.vk_commitments(circ, srs), 0x1234567
    proof = pp.create_proof(circ, srs, adv, inst, vk_repr, bytes(range(16)), multiopen)
    s_g2 = pr.ec_mul(pr.G2_GEN, S_SECRET)
    assert pv.verify(circ, vk_points, vk_repr, inst, proof, s_g2, multiopen=multiopen)
    bad = bytearray(proof)
    bad[40] ^= 1
    try:
        assert not pv.verify(circ, vk_points, vk_repr, inst, bytes(bad), s_g2, multiopen=multiopen)
    except AssertionError:
        pass                                  # the verifier may also refuse a malformed encoding outright
    # another seed: other blinding, other bytes, still accepted
    proof2 = pp.create_proof(circ, srs, adv, inst, vk_repr, bytes(16), multiopen)
    assert proof2 != proof and pv.verify(circ, vk_points, vk_repr, inst, proof2, s_g2, multiopen=multiopen)


def test_oracle_prover_many_rotations():
    circ, adv, inst = build_rotation_circuit(6, seed=4, window=6, blinding_factors=10)
    srs = pp.Srs(circ.k, S_SECRET)
    vk_points = pp.vk_commitments(circ, srs)
    proof = pp.create_proof(circ, srs, adv, inst, 99, bytes(16), "shplonk")
    assert pv.verify(circ, vk_points, 99, inst, proof, pr.ec_mul(pr.G2_GEN, S_SECRET), multiopen="shplonk")


def test_instance_slices_are_absorbed_as_given():
    """halo2 absorbs exactly the instance values it is handed (a circuit with 8 public inputs
    feeds 8 scalars to the transcript), pads the column with zeros and refuses more values than
    usable rows.  The slice proof differs from the proof over the n-row column image (other
    transcript) and each verifies only with the instance form it was made for."""
    circ, adv, inst = build_circuit(5, seed=3, wide=False)
    assert not any(inst[0][8:])
    short = [inst[0][:8]]
    srs = pp.Srs(circ.k, S_SECRET)
    vk_points, vk_repr = pp.vk_commitments(circ, srs), 0x7654321
    s_g2 = pr.ec_mul(pr.G2_GEN, S_SECRET)
    p_short = pp.create_proof(circ, srs, adv, short, vk_repr, bytes(range(16)), "shplonk")
    p_full = pp.create_proof(circ, srs, adv, inst, vk_repr, bytes(range(16)), "shplonk")
    assert p_short != p_full
    assert pv.verify(circ, vk_points, vk_repr, short, p_short, s_g2, multiopen="shplonk")
    assert pv.verify(circ, vk_points, vk_repr, inst, p_full, s_g2, multiopen="shplonk")
    for proof, ins in ((p_short, inst), (p_full, short)):
        try:
            assert not pv.verify(circ, vk_points, vk_repr, ins, proof, s_g2, multiopen="shplonk")
        except AssertionError:
            pass
    with pytest.raises(ValueError, match="InstanceTooLarge"):
        pp.create_proof(circ, srs, adv, [inst[0][:circ.u + 1]], vk_repr, bytes(16), "gwc")


@pytest.mark.parametrize("n_inputs,input_degree,gate_degree,multiopen", [(1, 2, 3, "shplonk"), (2, 1, 3, "shplonk"), (3, 1, 5, "gwc"), (5, 1, 9, "shplonk")])
def test_merged_and_chunked_lookups(n_inputs, input_degree, gate_degree, multiopen):
    """mv-lookup arguments with several input tuples per table (halo2 `chunk_lookups`): 1, 2 and 5
    input sets in one argument, a table whose inputs overflow into a second argument, a second
    table, duplicate table rows; the first case has no gate above degree 3, so the circuit degree
    comes from the lookup alone (required_degree)."""
    from plonk_fixtures import build_multi_lookup_circuit
    circ, adv, inst = build_multi_lookup_circuit(5, seed=n_inputs, n_inputs=n_inputs, input_degree=input_degree, gate_degree=gate_degree)
    shapes = sorted(len(lk.inputs) for lk in circ.lookups)
    want = {(1, 2, 3): (5, [1, 1]), (2, 1, 3): (5, [1, 2]), (3, 1, 5): (5, [1, 1, 2]), (5, 1, 9): (9, [1, 5])}[(n_inputs, input_degree, gate_degree)]
    assert (circ.degree(), shapes) == want
    assert pv.check_witness(circ, adv, inst) is None
    srs = pp.Srs(circ.k, S_SECRET)
    vk_points = pp.vk_commitments(circ, srs)
    vk_repr = pv.default_vk_repr(circ, vk_points)
    s_g2 = pr.ec_mul(pr.G2_GEN, S_SECRET)
    proof = pp.create_proof(circ, srs, adv, inst, vk_repr, bytes(range(16)), multiopen)
    assert pv.verify(circ, vk_points, vk_repr, inst, proof, s_g2, multiopen=multiopen)
    # an input that is not in the table cannot be proved
    bad = [list(col) for col in adv]
    row = next(r for r in range(circ.u) if circ.fixed[0][r] == 1)
    bad[1][row] = (bad[1][row] + 1) % pv.R
    assert pv.check_witness(circ, bad, inst) is not None
    with pytest.raises(AssertionError):
        pp.create_proof(circ, srs, bad, inst, vk_repr, bytes(16), multiopen)


@pytest.mark.parametrize("kind", ["poseidon", "evm"])
def test_poseidon_and_evm_transcripts(kind):
    """gen_snark_shplonk's Poseidon transcript and gen_evm_proof_shplonk's Keccak transcript
    [REF prover/src/common/prover/utils.rs:31], [REF prover/src/common/prover/evm.rs:67]"""
    circ, adv, inst = build_circuit(5, seed=6, wide=False)
    short = [inst[0][:8]]
    srs = pp.Srs(circ.k, S_SECRET)
    vk_points, vk_repr = pp.vk_commitments(circ, srs), 0xABCDEF
    s_g2 = pr.ec_mul(pr.G2_GEN, S_SECRET)
    proof = pp.create_proof(circ, srs, adv, short, vk_repr, bytes(range(16)), "shplonk", transcript=kind)
    assert pv.verify(circ, vk_points, vk_repr, short, proof, s_g2, multiopen="shplonk", transcript=kind)
    other = "evm" if kind == "poseidon" else "poseidon"
    try:
        assert not pv.verify(circ, vk_points, vk_repr, short, proof, s_g2, multiopen="shplonk", transcript=other)
    except AssertionError:
        pass
