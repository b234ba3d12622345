"""GPU: the configuration `bench.py` measures, under pytest -- so that the driver's own GPU run vouches for the headline and not only
the bench's `verified_by_oracle`.  SuperCircuit shape at k = 20 (BASELINE configs[3] stand-in, SURVEY 8d config 4: 1000 advice / 150
fixed / 150 permutation columns, 100 lookups, degree 9), witness cells ~60 % zero / ~30 % below 2^16 / 10 % uniform, THREE advice phases
with the SuperCircuit's challenge structure [REF zkevm-circuits/src/util.rs:120-133], SHPLONK + Blake2b as at
[REF circuit-benchmarks/src/super_circuit.rs:117-132], witness RESIDENT on the device and handed over IN PLACE through
zk_proof_advice_phase_dev, run-end and first-difference commitments on.  Required of the proof:
  * accepted by oracle/plonk_verifier.verify (the verifier that accepts the reference's own ChunkProof), rejected with one bit flipped;
  * the same bytes as the session fed HOST columns through zk_proof_advice_phase;
  * the same bytes with the structure-reading commitment paths off (ZK_MSM_RUNS=0 ZK_MSM_DIFF=0).
A box without the host memory for the witness FAILS (a skipped headline-size test reads as green) unless ZK_ALLOW_SMALL_HOST=1."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
S = 0x5EC2E7
SHAPE = (20, 1000, 150, 150, 100, 9)


def require_host_memory(gib: int):
    import psutil
    if psutil.virtual_memory().available >= (gib << 30):
        return
    if os.environ.get("ZK_ALLOW_SMALL_HOST") == "1":
        pytest.skip(f"less than {gib} GiB of host memory available (ZK_ALLOW_SMALL_HOST=1)")
    pytest.fail(f"less than {gib} GiB of host memory available: the headline-size test cannot run here (set ZK_ALLOW_SMALL_HOST=1 to skip it knowingly)")


def test_the_benched_configuration_at_k20(ctx, cref):
    require_host_memory(64)
    import bench_proof as bp
    from oracle import pairing as pr, plonk_verifier as pv

    circ, blob, adv_m, inst_m, inst, rlc = bp.build_shape(ctx, *SHAPE, dist="survey", phases=True)
    assert (circ.k, circ.A, circ.F, len(circ.perm_cols), len(circ.lookups), circ.degree()) == SHAPE
    assert circ.num_phases() == 3 and len(circ.challenge_phase) == 3
    npub = [int(np.flatnonzero(np.asarray(a).reshape(-1, 4).any(axis=1))[-1]) + 1 if np.asarray(a).any() else 0 for a in inst_m]
    inst = [list(col[:m]) for col, m in zip(inst, npub)]
    inst_m = [np.ascontiguousarray(a[:m]) for a, m in zip(inst_m, npub)]
    srs = ctx.srs_setup_with_s(circ.k, cref.fr_const(S))
    pk = ctx.pk_create(srs, blob)
    del blob
    adv_dev = [ctx.to_device(a) for a in adv_m]
    driver = bp.PhaseDriver(ctx, circ, adv_dev, rlc)
    try:
        com, rep = pk.vk(circ.F + len(circ.perm_cols))
        vk_points, vk_repr = cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0]

        def resident():
            sess = ctx.proof_session(pk, inst_m, bytes(16), instance_slices=True)
            sess.set_multiopen(1)
            driver.run(sess)
            return sess.finish()

        def verify(proof):
            try:
                return bool(pv.verify(circ, vk_points, vk_repr, inst, proof, pr.ec_mul(pr.G2_GEN, S), multiopen="shplonk"))
            except AssertionError:       # malformed point encodings
                return False

        proof = resident()
        assert len(proof) > 40000
        assert verify(proof)
        bad = bytearray(proof)
        bad[len(bad) // 3] ^= 4
        assert not verify(bytes(bad))
        # the structure-reading commitment paths off: the same bytes
        os.environ["ZK_MSM_RUNS"], os.environ["ZK_MSM_DIFF"] = "0", "0"
        try:
            assert resident() == proof
        finally:
            os.environ.pop("ZK_MSM_RUNS", None)
            os.environ.pop("ZK_MSM_DIFF", None)
        # host columns through zk_proof_advice_phase (what a Rust caller of create_proof holds): the same bytes
        ph, A = circ.advice_phase, circ.A
        sess = ctx.proof_session(pk, inst_m, bytes(16), instance_slices=True)
        sess.set_multiopen(1)
        ch0 = sess.advice_phase({i: adv_m[i] for i in range(A - 2) if ph[i] == 0})
        driver._rlc(0, ch0[0], rlc["w"])
        w_h = adv_dev[rlc["w"]].download((circ.n, 4))
        ch1 = sess.advice_phase({**{i: adv_m[i] for i in range(A - 2) if ph[i] == 1}, rlc["w"]: w_h})
        driver._rlc(rlc["w"], ch1[0], rlc["t"])
        t_h = adv_dev[rlc["t"]].download((circ.n, 4))
        sess.advice_phase({**{i: adv_m[i] for i in range(A - 2) if ph[i] == 2}, rlc["t"]: t_h})
        assert sess.finish() == proof
    finally:
        driver.free()
        for b_ in adv_dev:
            b_.free()
        pk.destroy()
        srs.destroy()
