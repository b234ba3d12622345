"""GPU: full proofs at BASELINE.json's sizes under pytest (configs 3 and 4), so that the driver's own GPU run vouches for them and
not only bench.py: the Keccak shape at k = 18 and the SuperCircuit shape at k = 20 (synthetic-shape stand-ins, SURVEY 8d: the
reference's witnesses need its Rust + Go toolchain), SHPLONK as at [REF circuit-benchmarks/src/super_circuit.rs:117-132], verified
by the oracle verifier that accepts the reference's own ChunkProof (tests/test_reference_chunk_proof.py); the Keccak shape at
k = 16 byte-equal to the restated CPU prover (oracle/cpu_prover.py).  Shapes: [REF circuit-benchmarks/src/packed_multi_keccak.rs:42-55],
[REF circuit-benchmarks/src/super_circuit.rs:45-98]."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
S = 0x5EC2E7


def _prove(ctx, cref, circ, blob, adv_m, inst_m, inst, transcript_kind=None, seed=bytes(16)):
    """keygen + one SHPLONK session over the instance slices; returns (proof, vk points, vk repr, instance slices)"""
    npub = [int(np.flatnonzero(np.asarray(a).reshape(-1, 4).any(axis=1))[-1]) + 1 if np.asarray(a).any() else 0 for a in inst_m]
    inst = [list(col[:m]) for col, m in zip(inst, npub)]
    inst_m = [np.ascontiguousarray(a[:m]) for a, m in zip(inst_m, npub)]
    srs = ctx.srs_setup_with_s(circ.k, cref.fr_const(S))
    pk = ctx.pk_create(srs, blob)
    try:
        com, rep = pk.vk(circ.F + len(circ.perm_cols))
        sess = ctx.proof_session(pk, inst_m, seed, instance_slices=True)
        sess.set_multiopen(1)
        if transcript_kind is not None:
            sess.set_transcript_kind(transcript_kind)
        sess.advice_phase({i: c for i, c in enumerate(adv_m)})
        proof = sess.finish()
    finally:
        pk.destroy()
        srs.destroy()
    return proof, cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0], inst


def _verify(circ, vk_points, vk_repr, inst, proof, transcript="blake2b"):
    from oracle import pairing as pr, plonk_verifier as pv
    try:
        return bool(pv.verify(circ, vk_points, vk_repr, inst, proof, pr.ec_mul(pr.G2_GEN, S), multiopen="shplonk", transcript=transcript))
    except AssertionError:           # malformed point encodings
        return False


def test_keccak_shape_k18_proof_is_accepted(ctx, cref):
    """BASELINE configs[2]: Keccak circuit k = 18 (stand-in: 96 advice columns, 59 unusable rows, 13-rotation gates, degree 9, 7 lookups)"""
    import bench_proof as bp
    circ, blob, adv_m, inst_m, inst = bp.build_keccak_shape(ctx, 18)
    assert circ.k == 18 and circ.degree() == 9 and circ.bf == 58
    proof, vk_points, vk_repr, inst = _prove(ctx, cref, circ, blob, adv_m, inst_m, inst)
    assert _verify(circ, vk_points, vk_repr, inst, proof)
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    assert not _verify(circ, vk_points, vk_repr, inst, bytes(bad))
    # and under the Poseidon transcript of gen_snark_shplonk [REF prover/src/common/prover/utils.rs:31]
    proof, vk_points, vk_repr, inst = _prove(ctx, cref, circ, blob, adv_m, inst_m, inst, transcript_kind=1)
    assert _verify(circ, vk_points, vk_repr, inst, proof, transcript="poseidon")


def test_keccak_shape_k16_bytes_equal_the_restated_cpu_prover(ctx, cref):
    """the same circuit, witness, seed and vk.transcript_repr through the GPU session and through halo2's create_proof restated over
    arrays on the host (whole-extended-domain evaluate_h): the same bytes"""
    import bench_proof as bp
    from oracle import cpu_prover as cp
    k = 16
    circ, blob, adv_m, inst_m, inst = bp.build_keccak_shape(ctx, k)
    proof, vk_points, vk_repr, inst_s = _prove(ctx, cref, circ, blob, adv_m, inst_m, inst)
    circ_h, adv_h, inst_h = bp.build_keccak_shape(None, k)
    assert inst_h == inst_s
    want = cp.create_proof(circ_h, cp.Srs(k, S), adv_h, inst_h, vk_repr, bytes(16), "shplonk", key=cp.keygen(circ_h))
    assert proof == want
    assert _verify(circ, vk_points, vk_repr, inst_s, proof)


def test_supercircuit_shape_k20_proof_is_accepted(ctx, cref):
    """BASELINE configs[3] on one GPU: k = 20, 1000 advice / 150 fixed / 150 permutation columns, 100 lookups, degree 9
    (SURVEY 8d config 4 stand-in).  Needs ~45 GiB of host memory for the witness and the key blob."""
    from test_gpu_headline_config import require_host_memory
    require_host_memory(64)          # fails (does not skip) on a small host unless ZK_ALLOW_SMALL_HOST=1
    import bench_proof as bp
    circ, blob, adv_m, inst_m, inst = bp.build_shape(ctx, 20, 1000, 150, 150, 100, 9)
    assert (circ.k, circ.A, circ.F, len(circ.perm_cols), len(circ.lookups), circ.degree()) == (20, 1000, 150, 150, 100, 9)
    proof, vk_points, vk_repr, inst = _prove(ctx, cref, circ, blob, adv_m, inst_m, inst)
    assert _verify(circ, vk_points, vk_repr, inst, proof)
    bad = bytearray(proof)
    bad[40] ^= 1
    assert not _verify(circ, vk_points, vk_repr, inst, bytes(bad))


def test_bundle_shape_k21_proof_is_accepted(ctx, cref):
    """BASELINE configs[4] stand-in at the size `bench.py` times it (`proof.bundle_shape_k21`): the halo2-base layout of
    [REF aggregator/configs/bundle_circuit.config] (degree 21, 5 + 1 advice, 1 fixed) under the Poseidon transcript of
    gen_snark_shplonk [REF prover/src/common/prover/recursion.rs:60-77]: accepted by the circuit-driven oracle verifier, rejected with
    one bit flipped -- so that the driver's own GPU run vouches for that line, not only the bench's `verified_by_oracle`."""
    import bench_proof as bp
    circ, blob, adv_m, inst_m, inst = bp.build_halo2_base_shape(ctx, 21, 5, 1, 20)
    assert circ.k == 21 and circ.A == 6 and circ.degree() == 5
    proof, vk_points, vk_repr, inst = _prove(ctx, cref, circ, blob, adv_m, inst_m, inst, transcript_kind=1)
    assert _verify(circ, vk_points, vk_repr, inst, proof, transcript="poseidon")
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    assert not _verify(circ, vk_points, vk_repr, inst, bytes(bad), transcript="poseidon")


def test_reference_constraint_system_at_k21_under_the_references_protocol(ctx, cref):
    """The constraint system of the reference-held ChunkProof at an aggregation layer's size (k = 21; the reference's own proof is
    k = 25), proved on the device under Poseidon + SHPLONK and handed to oracle/snark_verifier.py -- snark-verifier's
    PlonkSuccinctVerifier + decider restated -- DRIVEN BY THE REFERENCE'S OWN PROTOCOL OBJECT (quotient numerator, query and evaluation
    lists as snark-verifier compiled them; only the domain, the seven key commitments, the initial state and the instance count are
    ours).  The k = 8 twin is tests/test_gpu_reference_protocol.py."""
    import json
    import bench_proof as bp
    from oracle import pairing as pr, snark_verifier as sv
    from test_gpu_reference_protocol import protocol_for
    fixture = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_chunk_proof.json")))
    circ, blob, adv_m, inst_m, inst = bp.build_reference_cs_shape(ctx, 21)
    proof, vk_points, vk_repr, inst = _prove(ctx, cref, circ, blob, adv_m, inst_m, inst, transcript_kind=1, seed=bytes(range(16)))
    assert len(inst[0]) == 2
    s_g2 = pr.ec_mul(pr.G2_GEN, S)
    prot = sv.Protocol(protocol_for(fixture["protocol"], circ, vk_points, vk_repr, len(inst[0])))
    assert len(proof) == 32 * (sum(prot.num_witness) + prot.quotient["num_chunk"] + len(prot.evaluations) + 2) == 896
    assert sv.verify_snark(prot, inst, proof, pr.G2_GEN, s_g2)
    for word in (0, 5, 13, 27):
        bad = bytearray(proof)
        bad[32 * word + 7] ^= 8
        assert not sv.verify_snark(prot, inst, bytes(bad), pr.G2_GEN, s_g2), word
    assert _verify(circ, vk_points, vk_repr, inst, proof, transcript="poseidon")
