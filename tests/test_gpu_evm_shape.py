"""GPU: the EVM-style shape (round 6) -- an execution-state machine whose >= 5 000 constraints are
q_usable * q_step * state_selector_s * (constraint * condition) of degree 5 .. 9 over ~160 step columns read at rotations 0 / 1 / 2
[REF zkevm-circuits/src/evm_circuit/execution.rs:832-851, util/constraint_builder.rs:33-34,322-341, param.rs:10], beside the triple
gates, wide lookups and three advice phases of the SuperCircuit stand-in (bench_proof.build_shape(evm=...)): the configuration
`bench.py` now headlines.
  * at k = 6 / 8 (tests/plonk_fixtures.build_evm_circuit) the GPU proof is byte-equal to the oracle's big-int prover, and to itself
    with the class-program compiler off (ZK_QUOTIENT_DAG=0), the degree classes off, the cost gate off;
  * at k = 20, the benched configuration itself: accepted by the oracle verifier, rejected with one bit flipped, byte-equal with the
    degree classes off (`extra.degree_blind` of the bench line) and with the compiler off."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import bn254 as b  # noqa: E402
from oracle import pairing as pr  # noqa: E402
from oracle import plonk_verifier as pv  # noqa: E402
from plonk_fixtures import build_evm_circuit  # noqa: E402
from zkevm_circuits_amd import plonk  # noqa: E402

pytestmark = pytest.mark.gpu
S_SECRET = 0x5EC2E7
KNOBS = ({"ZK_QUOTIENT_DAG": "0"}, {"ZK_QUOTIENT_SPLIT": "0", "ZK_QUOTIENT_ADDSPLIT": "0"}, {"ZK_QUOTIENT_COSTGATE": "0"}, {"ZK_QUOTIENT_COSTGATE": "0", "ZK_QUOTIENT_DAG": "0"},
         {"ZK_QUOTIENT_DAG": "0", "ZK_QUOTIENT_GROUP": "0"}, {"ZK_QUOTIENT_KERNEL": "1"})


class env:
    def __init__(self, kv): self.kv = kv
    def __enter__(self): os.environ.update(self.kv)
    def __exit__(self, *a):
        for k_ in self.kv:
            os.environ.pop(k_, None)


@pytest.mark.parametrize("k,states,per_state", [(6, 6, 16), (8, 12, 24)])
def test_small_evm_shape_equals_the_oracle_prover(ctx, cref, k, states, per_state):
    from oracle import plonk_prover as pp
    circ, adv, inst = build_evm_circuit(k, seed=k, states=states, per_state=per_state)
    assert circ.degree() == 9 and min(g.degree() for g in circ.gates) >= 3
    assert sum(1 for g in circ.gates if 5 <= g.degree() <= 9) >= states * per_state
    assert pv.check_witness(circ, adv, inst) is None
    srs = ctx.srs_setup_with_s(k, cref.fr_const(S_SECRET))
    pk = ctx.pk_create(srs, circ.blob())
    seed = bytes(range(3, 19))
    try:
        com, rep = pk.vk(circ.F + len(circ.perm_cols))
        vk_points, vk_repr = cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0]

        def prove():
            sess = ctx.proof_session(pk, [plonk.column_to_mont(c) for c in inst], seed)
            sess.set_multiopen(1)
            sess.advice_phase({i: plonk.column_to_mont(c) for i, c in enumerate(adv)})
            return sess.finish()
        proof = prove()
        want = pp.create_proof(circ, pp.Srs(k, S_SECRET), adv, inst, vk_repr, seed, "shplonk")
        first_diff = next((i for i, (x, y) in enumerate(zip(proof, want)) if x != y), None)
        assert len(proof) == len(want) and first_diff is None, f"proofs differ from byte {first_diff}"
        assert pv.verify(circ, vk_points, vk_repr, inst, proof, pr.ec_mul(pr.G2_GEN, S_SECRET), multiopen="shplonk")
        for kv in KNOBS:
            with env(kv):
                assert prove() == proof, kv
        # an unsatisfied witness: a rw_counter cell off by one (the state-transition constraints of two steps fail; no condition guards them)
        bad = [list(c) for c in adv]
        bad[circ.evm_spec["ctr"]][8] = (bad[circ.evm_spec["ctr"]][8] + 1) % b.R_MOD
        assert pv.check_witness(circ, bad, inst) is not None
        sess = ctx.proof_session(pk, [plonk.column_to_mont(c) for c in inst], seed)
        sess.set_multiopen(1)
        sess.advice_phase({i: plonk.column_to_mont(c) for i, c in enumerate(bad)})
        assert not pv.verify(circ, vk_points, vk_repr, inst, sess.finish(), pr.ec_mul(pr.G2_GEN, S_SECRET), multiopen="shplonk")
    finally:
        pk.destroy()
        srs.destroy()


def test_plan_of_a_key_is_reported(ctx, cref):
    """zk_pk_quotient_plan: what the bench line reports about the evaluator's program (classes, instructions, parking slots)"""
    circ, adv, inst = build_evm_circuit(6, seed=2)
    srs = ctx.srs_setup_with_s(6, cref.fr_const(S_SECRET))
    pk = ctx.pk_create(srs, circ.blob())
    try:
        plan = pk.quotient_plan()
        assert plan["constraints"] == len(circ.gates) + 2 + 1 + 3 and plan["expression_graph"] == 1      # gates + permutation (l0, l_last, one chunk) + one lookup (3)
        assert sum(c["instructions"] for c in plan["classes"] if c["used"]) > 100
        assert max(c["slots_alive"] for c in plan["classes"]) <= 64
    finally:
        pk.destroy()
        srs.destroy()


def test_the_benched_evm_configuration_at_k20(ctx, cref):
    from test_gpu_headline_config import require_host_memory
    require_host_memory(64)
    import bench_proof as bp
    shape = (20, 1000, 150, 150, 100, 9)
    circ, blob, adv_m, inst_m, inst, rlc = bp.build_shape(ctx, *shape, dist="survey", phases=True, evm=dict(bp.EVM_DEFAULT))
    assert (circ.k, circ.A, circ.F, len(circ.perm_cols), len(circ.lookups), circ.degree()) == shape
    assert sum(1 for g in circ.gates if 5 <= g.degree() <= 9) >= 5000
    assert {len(lk.table) for lk in circ.lookups} == {4, 6, 8}
    npub = [int(np.flatnonzero(np.asarray(a).reshape(-1, 4).any(axis=1))[-1]) + 1 if np.asarray(a).any() else 0 for a in inst_m]
    inst = [list(col[:m]) for col, m in zip(inst, npub)]
    inst_m = [np.ascontiguousarray(a[:m]) for a, m in zip(inst_m, npub)]
    srs = ctx.srs_setup_with_s(circ.k, cref.fr_const(S_SECRET))
    pk = ctx.pk_create(srs, blob)
    del blob
    adv_dev = [ctx.to_device(a) for a in adv_m]
    driver = bp.PhaseDriver(ctx, circ, adv_dev, rlc)
    try:
        plan = pk.quotient_plan()
        assert sum(c["instructions"] for c in plan["classes"] if c["used"]) >= 50000
        assert max(c["slots_alive"] for c in plan["classes"]) <= 64
        com, rep = pk.vk(circ.F + len(circ.perm_cols))
        vk_points, vk_repr = cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0]

        def resident():
            sess = ctx.proof_session(pk, inst_m, bytes(16), instance_slices=True)
            sess.set_multiopen(1)
            driver.run(sess)
            return sess.finish()

        def verify(proof):
            try:
                return bool(pv.verify(circ, vk_points, vk_repr, inst, proof, pr.ec_mul(pr.G2_GEN, S_SECRET), multiopen="shplonk"))
            except AssertionError:
                return False
        proof = resident()
        assert verify(proof)
        bad = bytearray(proof)
        bad[len(bad) // 3] ^= 4
        assert not verify(bytes(bad))
        with env({"ZK_QUOTIENT_SPLIT": "0", "ZK_QUOTIENT_ADDSPLIT": "0"}):       # degree classes off: every column on all 8 cosets, one class
            assert resident() == proof
        with env({"ZK_QUOTIENT_DAG": "0"}):                                        # the exported trees, as round 5 assembled them
            assert resident() == proof
        with env({"ZK_QUOTIENT_SLICES": "0"}):                                     # class programs in one piece (by default the large ones are cut into slices that share their rows' operands)
            assert resident() == proof
        with env({"ZK_QUOTIENT_KERNEL": "1"}):                                     # round 5's interpreter
            assert resident() == proof
    finally:
        driver.free()
        for b_ in adv_dev:
            b_.free()
        pk.destroy()
        srs.destroy()
