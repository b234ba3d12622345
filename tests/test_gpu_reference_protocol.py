"""GPU: a proof of THIS prover for the constraint system of the reference-held ChunkProof, verified by the protocol-driven oracle
(`oracle/snark_verifier.py`, snark-verifier's PlonkSuccinctVerifier + decider restated) USING THE REFERENCE'S OWN PROTOCOL OBJECT.

`tests/golden/reference_chunk_proof.json: protocol` is the serde image of the `PlonkProtocol` snark-verifier compiled for a real
halo2-base circuit (`aggregator/data/batch-task.json`): its quotient numerator, query list and evaluation list say what a proof of
that constraint system must contain, in which order.  Here the same constraint system is built at k = 8 with a witness of our
own, proved on the device under the Poseidon transcript + SHPLONK (what gen_snark_shplonk instantiates
[REF prover/src/common/prover/utils.rs:31]), and handed to that verifier with the protocol's circuit-independent members left as
the reference wrote them -- only the domain, the seven key commitments, `transcript_initial_state` and the instance count are ours.
The verifier accepts the reference's proof (tests/test_reference_chunk_proof.py); it must accept ours, and reject it bit-flipped."""
import copy
import json
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import bn254 as b  # noqa: E402
from oracle import pairing as pr  # noqa: E402
from oracle import plonk_verifier as pv  # noqa: E402
from oracle import snark_verifier as sv  # noqa: E402
from zkevm_circuits_amd import binding, plonk  # noqa: E402

pytestmark = pytest.mark.gpu
R = plonk.R_MOD
S_SECRET = 0x5EC2E7
HERE = os.path.dirname(os.path.abspath(__file__))


def limbs4(v: int):
    return [(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]


def mont_limbs(v: int, mod: int):
    return limbs4((v << 256) % mod)


def build_reference_cs(k: int, seed: int = 1):
    """The fixture's constraint system: fixed 0 = lookup table, 1 = constants (equality-enabled), 2 = q_gate, 3 = q_lookup; one advice
    column a with the gate q_gate * (a + a.rot(1) * a.rot(2) - a.rot(3)); the lookup (q_lookup * a) in table; permutation over
    (fixed 1, advice 0, instance 0); blinding_factors = 6.  Queries registered in the order of the fixture's evaluation list."""
    circ = plonk.Circuit(k, num_fixed=4, num_advice=1, num_instance=1, blinding_factors=6)
    table, consts, q_gate, q_lookup = (circ.fixed_col(i) for i in range(4))
    a = circ.advice_col(0)
    circ.enable_equality(plonk.FIXED, 1)                        # fixed query (1, 0) first, as in the fixture
    circ.add_gate(q_gate * (a + a.rot(1) * a.rot(2) - a.rot(3)))
    circ.lookup_any("range", [q_lookup * a], [table])
    circ.chunk_lookups()
    circ.enable_equality(plonk.ADVICE, 0)
    circ.enable_equality(plonk.INSTANCE, 0)
    # the fixture's evaluation order of the fixed columns: 1, 0, 2, 3
    circ.fixed_queries = [(1, 0), (0, 0), (2, 0), (3, 0)]
    assert circ.advice_queries == [(0, 0), (0, 1), (0, 2), (0, 3)] and circ.perm_cols == [(plonk.FIXED, 1), (plonk.ADVICE, 0), (plonk.INSTANCE, 0)]
    assert circ.degree() == 5 and circ.halo2_blinding_factors() == 6
    n, u = circ.n, circ.u
    rng = random.Random(seed)
    tab_n = 64
    for i in range(tab_n):
        circ.fixed[0][i] = i
    adv = [0] * n
    gates = list(range(0, u - 40, 4))
    for r0 in gates:                                            # disjoint gate instances a, b, c, d = a + b c
        x, y, z = (rng.randrange(R) for _ in range(3))
        adv[r0], adv[r0 + 1], adv[r0 + 2], adv[r0 + 3] = x, y, z, (x + y * z) % R
        circ.fixed[2][r0] = 1
    lk_rows = list(range(u - 36, u - 4))                        # range-checked cells below the gates
    for r_ in lk_rows:
        adv[r_] = rng.randrange(tab_n)
        circ.fixed[3][r_] = 1
    # constants: two advice cells are constrained to constants of fixed column 1; two gate outputs are public
    circ.fixed[1][0], circ.fixed[1][1] = adv[lk_rows[0]], adv[lk_rows[1]]
    circ.copy((plonk.FIXED, 1, 0), (plonk.ADVICE, 0, lk_rows[0]))
    circ.copy((plonk.FIXED, 1, 1), (plonk.ADVICE, 0, lk_rows[1]))
    circ.copy((plonk.ADVICE, 0, lk_rows[2]), (plonk.ADVICE, 0, lk_rows[3]))
    adv[lk_rows[3]] = adv[lk_rows[2]]
    inst = [adv[gates[0] + 3], adv[gates[1] + 3]]
    circ.copy((plonk.ADVICE, 0, gates[0] + 3), (plonk.INSTANCE, 0, 0))
    circ.copy((plonk.ADVICE, 0, gates[1] + 3), (plonk.INSTANCE, 0, 1))
    return circ, [adv], [inst]


def protocol_for(fixture_protocol: dict, circ, vk_points, vk_repr: int, num_instance: int) -> dict:
    """the reference's protocol object with the circuit-dependent members replaced (domain, key commitments, initial state, instance count)"""
    p = copy.deepcopy(fixture_protocol)
    omega = b.omega_for_k(circ.k)
    p["domain"] = {"k": circ.k, "n": circ.n, "n_inv": mont_limbs(pow(circ.n, -1, R), R), "gen": mont_limbs(omega, R), "gen_inv": mont_limbs(pow(omega, -1, R), R)}
    p["preprocessed"] = [{"x": mont_limbs(pt[0], b.P_MOD), "y": mont_limbs(pt[1], b.P_MOD)} for pt in vk_points]
    p["transcript_initial_state"] = mont_limbs(vk_repr, R)
    p["num_instance"] = [num_instance]
    p["accumulator_indices"] = []
    return p


def test_our_proof_under_the_references_protocol(ctx, cref):
    fixture = json.load(open(os.path.join(HERE, "golden", "reference_chunk_proof.json")))
    k = 8
    circ, adv, inst = build_reference_cs(k)
    full_inst = [inst[0] + [0] * (circ.n - len(inst[0]))]
    assert pv.check_witness(circ, adv, full_inst) is None
    srs = ctx.srs_setup_with_s(k, cref.fr_const(S_SECRET))
    pk = ctx.pk_create(srs, circ.blob())
    try:
        com, rep = pk.vk(circ.F + len(circ.perm_cols))
        vk_points, vk_repr = cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0]
        sess = ctx.proof_session(pk, [plonk.column_to_mont(inst[0])], bytes(range(16)), instance_slices=True)
        sess.set_multiopen(1)
        sess.set_transcript_kind(binding.TRANSCRIPT_POSEIDON)
        sess.advice_phase({0: plonk.column_to_mont(adv[0])})
        proof = sess.finish()
    finally:
        pk.destroy()
        srs.destroy()
    s_g2 = pr.ec_mul(pr.G2_GEN, S_SECRET)
    # the layout the reference's protocol prescribes: 1 + 1 + 3 witness points, 4 quotient pieces, 17 evaluations, 2 opening points
    prot = sv.Protocol(protocol_for(fixture["protocol"], circ, vk_points, vk_repr, len(inst[0])))
    assert len(proof) == 32 * (sum(prot.num_witness) + prot.quotient["num_chunk"] + len(prot.evaluations) + 2) == 896
    assert sv.verify_snark(prot, inst, proof, pr.G2_GEN, s_g2)
    for word in (0, 3, 7, 12, 20, 26, 27):
        bad = bytearray(proof)
        bad[32 * word + 5] ^= 2
        assert not sv.verify_snark(prot, inst, bytes(bad), pr.G2_GEN, s_g2), word
    assert not sv.verify_snark(prot, [[inst[0][0], (inst[0][1] + 1) % R]], proof, pr.G2_GEN, s_g2)
    # the halo2-style verifier (circuit-driven) agrees
    assert pv.verify(circ, vk_points, vk_repr, inst, proof, s_g2, multiopen="shplonk", transcript="poseidon")
    # and the random polynomial's slot holds g[0] with evaluation 1, as in the reference's proof
    g0 = cref.affine_from_mont(np.asarray(ctx.srs_setup_with_s(k, cref.fr_const(S_SECRET)).download_g()[:1]))[0]
    assert b.g1_decompress(proof[32 * 4:32 * 5]) == g0 == b.G1_GEN
    assert int.from_bytes(proof[32 * 17:32 * 18], "little") == 1
