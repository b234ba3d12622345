"""GPU: full proofs (keygen_pk + create_proof with the GWC multi-open on the MI355X) accepted by
the oracle's pairing-based verifier -- the reference's own acceptance criterion for this path is
"verify_proof accepts" [REF circuit-benchmarks/src/super_circuit.rs:141-154]; soundness side:
tampered proofs / instances must be rejected, unsatisfied witnesses must be refused."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import bn254 as b  # noqa: E402
from oracle import pairing as pr  # noqa: E402
from oracle import plonk_verifier as pv  # noqa: E402
from plonk_fixtures import build_circuit  # noqa: E402
from zkevm_circuits_amd import plonk  # noqa: E402

pytestmark = pytest.mark.gpu
S_SECRET = 0x5EC2E7


def _prove(ctx, cref, circ, adv, inst, srs_by_k, seed=bytes(16)):
    pk = ctx.pk_create(srs_by_k[circ.k], circ.blob())
    ncom = circ.F + len(circ.perm_cols)
    com, rep = pk.vk(ncom)
    try:
        proof = ctx.create_proof(pk, [plonk.column_to_mont(c) for c in adv], [plonk.column_to_mont(c) for c in inst], seed)
    finally:
        pk.destroy()
    vk_points = cref.affine_from_mont(com)
    vk_repr = cref.from_mont(rep.reshape(1, 4))[0]
    return proof, vk_points, vk_repr


@pytest.fixture(scope="module")
def srs8(ctx, cref):
    """k -> SRS of exactly that size (commit_lagrange needs the Lagrange basis of the circuit's
    own domain; halo2 does the same through ParamsKZG::downsize)."""
    cache = {}

    class PerK(dict):
        def __missing__(self, k):
            self[k] = ctx.srs_setup_with_s(k, cref.fr_const(S_SECRET))
            return self[k]
    cache = PerK()
    yield cache
    for s in cache.values():
        s.destroy()


@pytest.fixture(scope="module")
def s_g2():
    return pr.ec_mul(pr.G2_GEN, S_SECRET)


@pytest.mark.parametrize("k,wide,seed", [(6, False, 1), (7, True, 2), (8, True, 3)])
def test_proof_verifies(ctx, cref, srs8, s_g2, k, wide, seed):
    circ, adv, inst = build_circuit(k, seed, wide)
    assert pv.check_witness(circ, adv, inst) is None
    proof, vk_points, vk_repr = _prove(ctx, cref, circ, adv, inst, srs8)
    assert pv.verify(circ, vk_points, vk_repr, inst, proof, s_g2)
    # deterministic for a fixed RNG seed, different for another seed (blinding), both valid
    proof2, _, _ = _prove(ctx, cref, circ, adv, inst, srs8)
    assert proof2 == proof
    proof3, _, _ = _prove(ctx, cref, circ, adv, inst, srs8, seed=bytes(range(16)))
    assert proof3 != proof and pv.verify(circ, vk_points, vk_repr, inst, proof3, s_g2)


def test_tampering_is_rejected(ctx, cref, srs8, s_g2):
    circ, adv, inst = build_circuit(6, 5, False)
    proof, vk_points, vk_repr = _prove(ctx, cref, circ, adv, inst, srs8)
    assert pv.verify(circ, vk_points, vk_repr, inst, proof, s_g2)
    d, P, L = circ.degree(), len(circ.perm_cols), len(circ.lookups)
    ncom = circ.A + 2 * L + 1 + (d - 1) + (P + d - 3) // (d - 2)
    off = 32 * ncom + 32 * 3          # a scalar inside the evaluation section
    bad = bytearray(proof)
    bad[off] ^= 1
    try:
        ok = pv.verify(circ, vk_points, vk_repr, inst, bytes(bad), s_g2)
    except AssertionError:
        ok = False
    assert not ok
    inst2 = [list(inst[0])]
    inst2[0][0] = (inst2[0][0] + 1) % b.R_MOD
    assert not pv.verify(circ, vk_points, vk_repr, inst2, proof, s_g2)
    assert not pv.verify(circ, vk_points, vk_repr, inst, proof, pr.ec_mul(pr.G2_GEN, S_SECRET + 1))


def test_unsatisfied_witness_is_refused_or_unverifiable(zk, ctx, cref, srs8, s_g2):
    circ, adv, inst = build_circuit(6, 7, False)
    assert pv.check_witness(circ, adv, inst) is None
    # break a copy constraint: the permutation argument cannot close
    a = circ.copies[0][0]
    adv_bad = [list(col) for col in adv]
    adv_bad[a[1]][a[2]] = (adv_bad[a[1]][a[2]] + 1) % b.R_MOD
    assert pv.check_witness(circ, adv_bad, inst) is not None
    with pytest.raises(zk.ZkError):
        _prove(ctx, cref, circ, adv_bad, inst, srs8)
    with pytest.raises(zk.ZkError):          # SRS of the wrong size is refused, not silently misused
        ctx.pk_create(srs8[7], circ.blob())
    # break a gate only: a proof comes out, but it must not verify
    copied = {(c_[0][1], c_[0][2]) for c_ in circ.copies if c_[0][0] == plonk.ADVICE} | {(c_[1][1], c_[1][2]) for c_ in circ.copies if c_[1][0] == plonk.ADVICE}
    row = next(r for r in range(circ.u) if circ.fixed[1][r] == 1 and (2, r + 1) not in copied)
    adv_bad = [list(col) for col in adv]
    adv_bad[2][row + 1] = (adv_bad[2][row + 1] + 1) % b.R_MOD
    assert pv.check_witness(circ, adv_bad, inst) is not None
    proof, vk_points, vk_repr = _prove(ctx, cref, circ, adv_bad, inst, srs8)
    assert not pv.verify(circ, vk_points, vk_repr, inst, proof, s_g2)
    # break a lookup: the prover refuses (input not in the table)
    row = next(r for r in range(circ.u) if circ.fixed[3][r] == 1)
    adv_bad = [list(col) for col in adv]
    adv_bad[1][row] = (adv_bad[1][row] + 1) % b.R_MOD
    with pytest.raises(zk.ZkError):
        _prove(ctx, cref, circ, adv_bad, inst, srs8)


@pytest.mark.parametrize("k,wide,seed", [(6, False, 11), (7, True, 12)])
def test_shplonk_multiopen(zk, ctx, cref, srs8, s_g2, k, wide, seed):
    """ProverSHPLONK (BDFG21) is what the reference's call sites instantiate
    [REF circuit-benchmarks/src/super_circuit.rs:117-132]: two commitments close the proof."""
    circ, adv, inst = build_circuit(k, seed, wide)
    pk = ctx.pk_create(srs8[k], circ.blob())
    com, rep = pk.vk(circ.F + len(circ.perm_cols))
    vk_points, vk_repr = cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0]

    def prove(kind):
        sess = ctx.proof_session(pk, [plonk.column_to_mont(c) for c in inst], bytes(16))
        sess.set_multiopen(kind)
        sess.advice_phase({i: plonk.column_to_mont(c) for i, c in enumerate(adv)})
        return sess.finish()
    gwc, shp = prove(0), prove(1)
    pk.destroy()
    def safe_verify(proof, scheme):
        try:
            return pv.verify(circ, vk_points, vk_repr, inst, proof, s_g2, multiopen=scheme)
        except AssertionError:      # malformed encoding / truncated proof
            return False
    assert len(shp) < len(gwc)                       # 2 closing commitments instead of one per point
    assert safe_verify(shp, "shplonk")
    assert safe_verify(gwc, "gwc")
    assert not safe_verify(shp, "gwc") and not safe_verify(gwc, "shplonk")
    bad = bytearray(shp)
    bad[-40] ^= 1            # inside the final commitment
    assert not safe_verify(bytes(bad), "shplonk")


def test_two_phase_proof_with_challenge(zk, ctx, cref, srs8, s_g2):
    """Phases as in the SuperCircuit [REF zkevm-circuits/src/util.rs:120-133]: columns a, b are
    first-phase; the RLC column c = a + r*b needs the challenge r squeezed after phase 0, so the
    host synthesises it between two zk_proof_advice_phase calls (what the Rust shim does with
    Circuit::synthesize per phase)."""
    import random
    k = 6
    circ = plonk.Circuit(k, num_fixed=1, num_advice=4, num_instance=0, blinding_factors=5)
    q, a, b_, c_, d_ = circ.fixed_col(0), circ.advice_col(0), circ.advice_col(1), circ.advice_col(2), circ.advice_col(3)
    circ.advice_phase = [0, 0, 1, 1]
    r = circ.challenge_usable_after(0)
    r2 = circ.challenge_usable_after(0)
    circ.add_gate(q * (a + r * b_ - c_))                 # RLC
    circ.add_gate(q * (c_ * r2 + a * a * b_ - d_))       # a second challenge, degree 4 with the selector
    circ.enable_equality(plonk.ADVICE, 2)
    rng = random.Random(3)
    n, u = circ.n, circ.u
    av, bv = [0] * n, [0] * n
    for row in range(u):
        circ.fixed[0][row] = 1
        av[row], bv[row] = rng.randrange(b.R_MOD), rng.randrange(b.R_MOD)
    pk = ctx.pk_create(srs8[k], circ.blob())
    com, rep = pk.vk(circ.F + len(circ.perm_cols))
    sess = ctx.proof_session(pk, [], bytes(16))
    ch = sess.advice_phase({0: plonk.column_to_mont(av), 1: plonk.column_to_mont(bv)})
    assert ch.shape == (2, 4)
    rv, r2v = cref.from_mont(ch)
    cv = [(av[i] + rv * bv[i]) % b.R_MOD if i < u else 0 for i in range(n)]
    dv = [(cv[i] * r2v + av[i] * av[i] * bv[i]) % b.R_MOD if i < u else 0 for i in range(n)]
    assert pv.check_witness(circ, [av, bv, cv, dv], [], challenges=[rv, r2v]) is None
    with pytest.raises(zk.ZkError):                      # wrong column set for phase 1
        sess.advice_phase({2: plonk.column_to_mont(cv)})
    assert sess.advice_phase({2: plonk.column_to_mont(cv), 3: plonk.column_to_mont(dv)}).shape == (0, 4)
    proof = sess.finish()
    pk.destroy()
    assert pv.verify(circ, cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0], [], proof, s_g2)
    # the oracle prover squeezes the same challenges after phase 0 and arrives at the same bytes
    from oracle import plonk_prover as pp
    assert proof == pp.create_proof(circ, pp.Srs(k, S_SECRET), [av, bv, cv, dv], [], cref.from_mont(rep.reshape(1, 4))[0], bytes(16), "gwc")
    # a witness built with the WRONG challenge must not verify
    pk = ctx.pk_create(srs8[k], circ.blob())
    sess = ctx.proof_session(pk, [], bytes(16))
    sess.advice_phase({0: plonk.column_to_mont(av), 1: plonk.column_to_mont(bv)})
    cbad = [(av[i] + (rv + 1) * bv[i]) % b.R_MOD if i < u else 0 for i in range(n)]
    sess.advice_phase({2: plonk.column_to_mont(cbad), 3: plonk.column_to_mont(dv)})
    bad = sess.finish()
    pk.destroy()
    assert not pv.verify(circ, cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0], [], bad, s_g2)


def _three_phase_circuit(k):
    """The SuperCircuit's phase structure [REF zkevm-circuits/src/util.rs:120-133]: `evm_word` and `keccak_input` usable after the
    FIRST phase, `lookup_input` after the SECOND; advice columns in each of the three phases.
      phase 0: a, b                          phase 1: w = a + evm_word * b,  kk = b + keccak_input * a   (the two RLCs)
      phase 2: t = w + lookup_input * kk,    s with s(row + 1) * w(row) = t(row)^2 * lookup_input on the rows q5 selects,
               lk = cells of w picked by copy constraints, looked up in w itself (a third-phase column feeding a lookup into a
               second-phase table, as the EVM circuit's lookup columns do [REF zkevm-circuits/src/evm_circuit/execution.rs:418-431])."""
    circ = plonk.Circuit(k, num_fixed=3, num_advice=7, num_instance=1, blinding_factors=5)
    q, q5, ql = circ.fixed_col(0), circ.fixed_col(1), circ.fixed_col(2)
    a, b_, w, kk, t, s_, lk = (circ.advice_col(i) for i in range(7))
    circ.advice_phase = [0, 0, 1, 1, 2, 2, 2]
    evm_word = circ.challenge_usable_after(0)
    keccak_input = circ.challenge_usable_after(0)
    lookup_input = circ.challenge_usable_after(1)
    circ.add_gate(q * (a + evm_word * b_ - w))
    circ.add_gate(q * (b_ + keccak_input * a - kk))
    circ.add_gate(q * (w + lookup_input * kk - t))
    circ.add_gate(q5 * (t * t * lookup_input - s_.rot(1) * w))
    circ.lookup_any("lk in w", [ql * lk], [w])
    circ.chunk_lookups()
    u = circ.u
    for row in range(u):
        circ.fixed[0][row] = 1
    for row in range(0, u - 3, 2):
        circ.fixed[1][row] = 1
    picks = [(3 * j + 1, (7 * j + 2) % (u - 1)) for j in range(6)]          # lk(row) == w(src)
    for row, src in picks:
        circ.fixed[2][row] = 1
        circ.copy((plonk.ADVICE, 6, row), (plonk.ADVICE, 2, src))
    circ.copy((plonk.ADVICE, 0, 0), (plonk.INSTANCE, 0, 0))
    return circ, picks


def test_three_phase_proof_as_the_supercircuit(zk, ctx, cref, srs8, s_g2):
    """Three zk_proof_advice_phase calls with challenges handed back twice -- the re-synthesis contract of
    [REF zkevm-circuits/src/super_circuit.rs:728-736] (`synthesize` runs once per phase; what needs `lookup_input` is assigned in
    the third).  Bytes equal the oracle prover's, which is handed the same per-phase synthesis as a callback; the oracle verifier
    (pinned by the reference's ChunkProof) accepts; a third phase synthesised with a wrong `lookup_input` is rejected."""
    import random
    from oracle import plonk_prover as pp
    k = 6
    circ, picks = _three_phase_circuit(k)
    n, u, R = circ.n, circ.u, b.R_MOD
    rng = random.Random(5)
    av = [rng.randrange(1, R) if i < u - 1 else 0 for i in range(n)]          # row u - 1: a = b = 0, so that w holds a zero for the unselected lookup rows
    bv = [rng.randrange(1, R) if i < u - 1 else 0 for i in range(n)]

    def synth(phase, ch):
        """what Circuit::synthesize assigns in `phase`, given the challenges squeezed so far"""
        if phase == 0:
            return {0: av, 1: bv}
        wv = [(av[i] + ch[0] * bv[i]) % R if i < u else 0 for i in range(n)]
        kv = [(bv[i] + ch[1] * av[i]) % R if i < u else 0 for i in range(n)]
        if phase == 1:
            return {2: wv, 3: kv}
        tv = [(wv[i] + ch[2] * kv[i]) % R if i < u else 0 for i in range(n)]
        sv, lv = [0] * n, [0] * n
        for row in range(u):
            if circ.fixed[1][row]:
                sv[row + 1] = tv[row] * tv[row] % R * ch[2] % R * b.fr_inv(wv[row]) % R
        for row, src in picks:
            lv[row] = wv[src]
        return {4: tv, 5: sv, 6: lv}
    inst = [[av[0]] + [0] * (n - 1)]
    pk = ctx.pk_create(srs8[k], circ.blob())
    com, rep = pk.vk(circ.F + len(circ.perm_cols))
    vk_points, vk_repr = cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0]

    def gpu_proof(tamper=False):
        sess = ctx.proof_session(pk, [plonk.column_to_mont(c) for c in inst], bytes(range(16)))
        sess.set_multiopen(1)
        ch = []
        got = sess.advice_phase({i: plonk.column_to_mont(c) for i, c in synth(0, ch).items()})
        assert got.shape == (2, 4)                       # evm_word, keccak_input
        ch += cref.from_mont(got)
        got = sess.advice_phase({i: plonk.column_to_mont(c) for i, c in synth(1, ch).items()})
        assert got.shape == (1, 4)                       # lookup_input
        ch += cref.from_mont(got)
        used = ch[:2] + [(ch[2] + 1) % R] if tamper else ch
        cols = synth(2, used)
        if tamper:
            cols[6] = synth(2, ch)[6]                    # keep the lookup and the copies satisfied: only the gates see the wrong challenge
        got = sess.advice_phase({i: plonk.column_to_mont(c) for i, c in cols.items()})
        assert got.shape == (0, 4)
        return sess.finish(), ch
    try:
        proof, ch = gpu_proof()
        bad, _ = gpu_proof(tamper=True)
    finally:
        pk.destroy()
    full = {}
    for ph in range(3):
        full.update(synth(ph, ch))
    assert pv.check_witness(circ, [full[i] for i in range(7)], inst, challenges=ch) is None
    assert pv.verify(circ, vk_points, vk_repr, inst, proof, s_g2, multiopen="shplonk")
    want = pp.create_proof(circ, pp.Srs(k, S_SECRET), [[0] * n for _ in range(7)], inst, vk_repr, bytes(range(16)), "shplonk", phase_witness=synth)
    assert proof == want
    assert pv.verify(circ, vk_points, vk_repr, inst, bad, s_g2, multiopen="shplonk") is False


@pytest.mark.parametrize("multiopen", ["gwc", "shplonk"])
def test_many_rotations_keccak_like(ctx, cref, srs8, s_g2, multiopen):
    """One column opened at 14 distinct rotations with 17 blinding rows (the Keccak circuit's query
    pattern): batched evaluations, rotation sets and witness polynomials of the
    multi-open argument all see large point sets."""
    from plonk_fixtures import build_rotation_circuit
    circ, adv, inst = build_rotation_circuit(7, seed=3)
    assert pv.check_witness(circ, adv, inst) is None
    pk = ctx.pk_create(srs8[circ.k], circ.blob())
    try:
        com, rep = pk.vk(circ.F + len(circ.perm_cols))
        sess = ctx.proof_session(pk, [plonk.column_to_mont(c) for c in inst], bytes(16))
        sess.set_multiopen(1 if multiopen == "shplonk" else 0)
        sess.advice_phase({i: plonk.column_to_mont(c) for i, c in enumerate(adv)})
        proof = sess.finish()
    finally:
        pk.destroy()
    vk_points, vk_repr = cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0]
    assert pv.verify(circ, vk_points, vk_repr, inst, proof, s_g2, multiopen=multiopen)
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    try:
        assert not pv.verify(circ, vk_points, vk_repr, inst, bytes(bad), s_g2, multiopen=multiopen)
    except AssertionError:
        pass


class _HostBlake2bWrite:
    """halo2's Blake2bWrite<_, G1Affine, Challenge255> kept on the host side of the ABI: what a Rust
    shim passes as `transcript` (here the same hash, so the bytes must equal the built-in ones)."""

    def __init__(self, cref):
        import hashlib
        self.cref = cref
        self.h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.proof = bytearray()

    def _point(self, raw64):
        import numpy as np
        return self.cref.affine_from_mont(np.frombuffer(raw64, dtype=np.uint64).reshape(1, 8))[0]

    def _scalar(self, raw32):
        import numpy as np
        return self.cref.from_mont(np.frombuffer(raw32, dtype=np.uint64).reshape(1, 4))[0]

    def common_point(self, raw64):
        pt = self._point(raw64)
        self.h.update(b"\x01" + (bytes(64) if pt is None else pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little")))

    def common_scalar(self, raw32):
        self.h.update(b"\x02" + self._scalar(raw32).to_bytes(32, "little"))

    def write_point(self, raw64):
        self.common_point(raw64)
        pt = self._point(raw64)
        from oracle import bn254
        self.proof += bn254.g1_compress(pt)             # x LE, parity of y in bit 254, identity = bit 255 (pinned by the reference's ChunkProof)

    def write_scalar(self, raw32):
        self.common_scalar(raw32)
        self.proof += self._scalar(raw32).to_bytes(32, "little")

    def squeeze_challenge(self):
        self.h.update(b"\x00")
        return self.cref.fr_const(b.fr_from_uniform_bytes(self.h.copy().digest())).tobytes()


@pytest.mark.parametrize("multiopen", [0, 1])
def test_external_transcript_matches_builtin(ctx, cref, srs8, multiopen):
    """zk_proof_set_transcript: with the host's transcript object doing Blake2b, the bytes it
    collects equal the proof of the built-in transcript (same session otherwise)."""
    circ, adv, inst = build_circuit(7, 4, True)
    pk = ctx.pk_create(srs8[circ.k], circ.blob())
    adv_m = {i: plonk.column_to_mont(c) for i, c in enumerate(adv)}
    inst_m = [plonk.column_to_mont(c) for c in inst]
    try:
        def run(external):
            sess = ctx.proof_session(pk, inst_m, bytes(range(16)))
            sess.set_multiopen(multiopen)
            tr = None
            if external:
                tr = _HostBlake2bWrite(cref)
                sess.set_transcript(tr)
            sess.advice_phase(adv_m)
            out = sess.finish()
            return bytes(tr.proof) if external else out, out
        builtin, _ = run(False)
        external, returned = run(True)
    finally:
        pk.destroy()
    assert returned == b""                    # the host transcript owns the proof
    assert len(builtin) > 500 and external == builtin


@pytest.mark.parametrize("k,wide,multiopen,vanishing", [(5, False, "gwc", "one"), (6, True, "shplonk", "one"), (6, True, "gwc", "uniform"), (6, True, "shplonk", "uniform")])
def test_proof_bytes_equal_the_oracle_prover(ctx, cref, srs8, k, wide, multiopen, vanishing):
    """Strongest parity statement for the whole path: for the same key, witness and seed the GPU
    session and the oracle's big-int restatement of create_proof produce the same bytes --
    every commitment, evaluation and opening agrees, not merely "the verifier accepts".  Both kinds of the
    vanishing argument's polynomial: the constant 1 of the reference's own proofs (default) and upstream's uniform one."""
    from oracle import plonk_prover as pp
    circ, adv, inst = build_circuit(k, seed=7, wide=wide)
    seed = bytes((3 * i + 1) & 0xFF for i in range(16))
    pk = ctx.pk_create(srs8[circ.k], circ.blob())
    try:
        com, rep = pk.vk(circ.F + len(circ.perm_cols))
        sess = ctx.proof_session(pk, [plonk.column_to_mont(c) for c in inst], seed)
        sess.set_multiopen(1 if multiopen == "shplonk" else 0)
        if vanishing == "uniform":
            sess.set_vanishing_random(0)
        sess.advice_phase({i: plonk.column_to_mont(c) for i, c in enumerate(adv)})
        gpu_proof = sess.finish()
    finally:
        pk.destroy()
    srs = pp.Srs(circ.k, S_SECRET)
    assert cref.affine_from_mont(com) == pp.vk_commitments(circ, srs)          # keygen agrees first
    vk_repr = cref.from_mont(rep.reshape(1, 4))[0]
    want = pp.create_proof(circ, srs, adv, inst, vk_repr, seed, multiopen, vanishing=vanishing)
    assert len(gpu_proof) == len(want)
    first_diff = next((i for i, (x, y) in enumerate(zip(gpu_proof, want)) if x != y), None)
    assert first_diff is None, f"proofs differ from byte {first_diff} (32-byte item {first_diff // 32})"


def test_device_resident_witness_gives_the_same_bytes(zk, ctx, cref, srs8):
    """zk_proof_advice_phase_dev: the witness columns handed over as device buffers (resident in HBM before the session starts)
    yield the bytes of the host-column call.  Copy mode leaves the caller's buffers as they were; in-place mode works in them and
    touches only their last blinding_factors + 1 rows; the same buffer given for two columns is refused there."""
    circ, adv, inst = build_circuit(7, seed=21, wide=True)
    pk = ctx.pk_create(srs8[circ.k], circ.blob())
    adv_m = [plonk.column_to_mont(c) for c in adv]
    inst_m = [plonk.column_to_mont(c) for c in inst]
    try:
        def run(mode):
            sess = ctx.proof_session(pk, inst_m, bytes(range(16)))
            sess.set_multiopen(1)
            if mode == "host":
                sess.advice_phase({i: c for i, c in enumerate(adv_m)})
                return sess.finish()
            bufs = {i: ctx.to_device(c) for i, c in enumerate(adv_m)}
            sess.advice_phase_dev(bufs, in_place=mode == "in_place")
            out = sess.finish()
            for i, b_ in bufs.items():
                got = b_.download((circ.n, 4))
                if mode == "copy":
                    assert np.array_equal(got, adv_m[i])
                else:
                    assert np.array_equal(got[:circ.u], adv_m[i][:circ.u]) and got[circ.u:].any()          # usable rows untouched, blinding rows written
                b_.free()
            return out
        host, copy, in_place = run("host"), run("copy"), run("in_place")
        sess = ctx.proof_session(pk, inst_m, bytes(range(16)))
        shared = ctx.to_device(adv_m[0])
        with pytest.raises(zk.ZkError, match="distinct"):
            sess.advice_phase_dev({i: shared for i in range(len(adv_m))}, in_place=True)
        sess.abort()
        shared.free()
    finally:
        pk.destroy()
    assert len(host) > 500 and host == copy == in_place


def test_rotation_proof_bytes_equal_the_oracle_prover(ctx, cref, srs8):
    from oracle import plonk_prover as pp
    from plonk_fixtures import build_rotation_circuit
    circ, adv, inst = build_rotation_circuit(6, seed=4, window=6, blinding_factors=10)
    pk = ctx.pk_create(srs8[circ.k], circ.blob())
    try:
        _, rep = pk.vk(circ.F + len(circ.perm_cols))
        sess = ctx.proof_session(pk, [plonk.column_to_mont(c) for c in inst], bytes(16))
        sess.set_multiopen(1)
        sess.advice_phase({i: plonk.column_to_mont(c) for i, c in enumerate(adv)})
        gpu_proof = sess.finish()
    finally:
        pk.destroy()
    want = pp.create_proof(circ, pp.Srs(circ.k, S_SECRET), adv, inst, cref.from_mont(rep.reshape(1, 4))[0], bytes(16), "shplonk")
    assert gpu_proof == want


@pytest.mark.parametrize("multiopen", ["gwc", "shplonk"])
def test_instance_slices_equal_the_oracle_prover(zk, ctx, cref, srs8, s_g2, multiopen):
    """zk_proof_begin_instances: halo2's instance slices as they are -- exactly the given values are
    absorbed, the column is zero-padded on the device.  Bytes equal the oracle prover's for the same
    slices, the oracle verifier accepts them with the slices (and not with the n-row image), and
    more values than usable rows is refused (Error::InstanceTooLarge)."""
    from oracle import plonk_prover as pp
    circ, adv, inst = build_circuit(6, seed=9, wide=True)
    assert not any(inst[0][8:])
    short = [inst[0][:8]]
    seed = bytes((5 * i + 2) & 0xFF for i in range(16))
    pk = ctx.pk_create(srs8[circ.k], circ.blob())
    try:
        com, rep = pk.vk(circ.F + len(circ.perm_cols))
        adv_m = {i: plonk.column_to_mont(c) for i, c in enumerate(adv)}
        sess = ctx.proof_session(pk, [plonk.column_to_mont(c) for c in short], seed, instance_slices=True)
        sess.set_multiopen(1 if multiopen == "shplonk" else 0)
        sess.advice_phase(adv_m)
        gpu_proof = sess.finish()
        with pytest.raises(zk.ZkError, match="InstanceTooLarge"):
            ctx.proof_session(pk, [plonk.column_to_mont(inst[0][:circ.u + 1])], seed, instance_slices=True)
        empty = ctx.proof_session(pk, [np.zeros((0, 4), dtype=np.uint64)], seed, instance_slices=True)   # no public input given at all
        empty.abort()
    finally:
        pk.destroy()
    vk_points, vk_repr = cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0]
    want = pp.create_proof(circ, pp.Srs(circ.k, S_SECRET), adv, short, vk_repr, seed, multiopen)
    assert gpu_proof == want
    assert pv.verify(circ, vk_points, vk_repr, short, gpu_proof, s_g2, multiopen=multiopen)
    try:
        assert not pv.verify(circ, vk_points, vk_repr, inst, gpu_proof, s_g2, multiopen=multiopen)
    except AssertionError:
        pass


def test_evm_circuit_sized_proof_k14(ctx, cref, s_g2):
    """BASELINE configs[0] is the EVM sub-circuit at k = 14 (the reference checks it with
    MockProver; [REF circuit-benchmarks/src/evm_circuit.rs:44-60]).  Its stand-in here: a synthetic
    circuit of that size class -- k = 14, 159 advice columns (the EVM circuit has 157), degree-5
    gates, a lookup, 160 permutation columns -- proved through the session API with the instance
    slice and accepted by the oracle's pairing verifier."""
    import bench_proof
    circ, blob, adv_m, inst_m, inst = bench_proof.build_large(ctx, 14, 53)
    assert circ.A == 159 and circ.k == 14
    npub = int(np.flatnonzero(inst_m[0].any(axis=1))[-1]) + 1
    srs = ctx.srs_setup_with_s(14, np.frombuffer(plonk.fr_mont_bytes(S_SECRET), dtype=np.uint64).copy())
    pk = ctx.pk_create(srs, blob)
    try:
        com, rep = pk.vk(circ.F + len(circ.perm_cols))
        sess = ctx.proof_session(pk, [np.ascontiguousarray(inst_m[0][:npub])], bytes(16), instance_slices=True)
        sess.set_multiopen(1)
        sess.advice_phase({i: c for i, c in enumerate(adv_m)})
        proof = sess.finish()
    finally:
        pk.destroy()
        srs.destroy()
    assert pv.verify(circ, cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0], [inst[0][:npub]], proof, s_g2, multiopen="shplonk")


def test_shared_subexpressions_through_intermediates(ctx, cref, srs8, s_g2):
    """The key blob exported with cross-gate common-subexpression elimination (TEE_TMP / PUSH_TMP,
    what halo2's GraphEvaluator does with its intermediates) gives the same proof bytes as the plain
    export and as the oracle prover, which evaluates every gate expression from scratch."""
    from oracle import plonk_prover as pp
    circ, adv, inst = build_circuit(6, seed=11, wide=True)
    q_mul, a, b_, cc = circ.fixed_col(0), circ.advice_col(0), circ.advice_col(1), circ.advice_col(2)
    circ.add_gate((q_mul * (a * b_ - cc)) * (a * b_ + 5))            # reuses a*b, a*b - c and the whole first gate
    circ.add_gate(q_mul * ((a * b_ - cc) * (a * b_ - cc)))
    progs = circ.compile_gates_cse()
    ops = [op for p in progs for op, _, _ in p]
    assert plonk.Q_TEE_TMP in ops and ops.count(plonk.Q_PUSH_TMP) >= 3
    assert sum(len(p) for p in progs) < sum(len(circ.compile(g)) for g in circ.gates)
    assert pv.check_witness(circ, adv, inst) is None
    seed = bytes((7 * i + 3) & 0xFF for i in range(16))
    proofs = []
    for cse in (False, True):
        pk = ctx.pk_create(srs8[circ.k], circ.blob(cse=cse))
        try:
            # two exports of the SAME circuit: the caller holds one vk.transcript_repr for both (the stand-in
            # the library derives hashes the exported programs, which differ)
            pk.set_transcript_repr(cref.to_mont([0x5EED5EED])[0])
            com, rep = pk.vk(circ.F + len(circ.perm_cols))
            sess = ctx.proof_session(pk, [plonk.column_to_mont(c) for c in inst], seed)
            sess.set_multiopen(1)
            sess.advice_phase({i: plonk.column_to_mont(c) for i, c in enumerate(adv)})
            proofs.append(sess.finish())
        finally:
            pk.destroy()
    assert proofs[0] == proofs[1]
    vk_points, vk_repr = cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0]
    want = pp.create_proof(circ, pp.Srs(circ.k, S_SECRET), adv, inst, vk_repr, seed, "shplonk")
    assert proofs[1] == want
    assert pv.verify(circ, vk_points, vk_repr, inst, proofs[1], s_g2, multiopen="shplonk")


def _session_proof(ctx, pk, adv, inst, seed, multiopen, transcript_kind=None, slices=False):
    sess = ctx.proof_session(pk, [plonk.column_to_mont(c) for c in inst], seed, instance_slices=slices)
    if transcript_kind is not None:
        sess.set_transcript_kind(transcript_kind)
    sess.set_multiopen(1 if multiopen == "shplonk" else 0)
    sess.advice_phase({i: plonk.column_to_mont(c) for i, c in enumerate(adv)})
    return sess.finish()


@pytest.mark.parametrize("n_inputs,input_degree,gate_degree,multiopen", [(1, 2, 3, "shplonk"), (2, 1, 3, "gwc"), (3, 1, 5, "shplonk"), (5, 1, 9, "shplonk")])
def test_merged_and_chunked_lookups_equal_the_oracle_prover(zk, ctx, cref, srs8, s_g2, n_inputs, input_degree, gate_degree, multiopen):
    """mv-lookup arguments as halo2's chunk_lookups() leaves them [REF zkevm-circuits/src/super_circuit/test.rs:59]:
    1, 2 and 5 input tuples in one argument, a table whose inputs overflow into a second argument, a
    second table, duplicate table rows.  The first shape has no gate above degree 3: the circuit
    degree (5) comes from the lookup's required_degree alone."""
    from oracle import plonk_prover as pp
    from plonk_fixtures import build_multi_lookup_circuit
    circ, adv, inst = build_multi_lookup_circuit(6, seed=10 + n_inputs, n_inputs=n_inputs, input_degree=input_degree, gate_degree=gate_degree)
    assert pv.check_witness(circ, adv, inst) is None
    seed = bytes(range(1, 17))
    pk = ctx.pk_create(srs8[circ.k], circ.blob())
    try:
        com, rep = pk.vk(circ.F + len(circ.perm_cols))
        shape = pk.shape()
        assert (shape["degree"], shape["L"], shape["advice_queries"]) == (circ.degree(), len(circ.lookups), len(circ.advice_queries))
        gpu_proof = _session_proof(ctx, pk, adv, inst, seed, multiopen)
        bad = [list(col) for col in adv]
        row = next(r for r in range(circ.u) if circ.fixed[0][r] == 1)      # an enabled row (the degree-2 inputs are gated by q)
        bad[1][row] = (bad[1][row] + 1) % b.R_MOD             # an input pair that is not in the table
        with pytest.raises(zk.ZkError, match="not in the table"):
            _session_proof(ctx, pk, bad, inst, seed, multiopen)
    finally:
        pk.destroy()
    vk_points, vk_repr = cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0]
    assert vk_repr == pv.default_vk_repr(circ, vk_points)
    want = pp.create_proof(circ, pp.Srs(circ.k, S_SECRET), adv, inst, vk_repr, seed, multiopen)
    first_diff = next((i for i, (x, y) in enumerate(zip(gpu_proof, want)) if x != y), None)
    assert len(gpu_proof) == len(want) and first_diff is None, f"proofs differ from byte {first_diff} (32-byte item {first_diff // 32 if first_diff is not None else -1})"
    assert pv.verify(circ, vk_points, vk_repr, inst, gpu_proof, s_g2, multiopen=multiopen)


def test_declared_degree_below_the_lookup_degree_is_refused(zk, ctx, srs8):
    """halo2's cs.degree() counts mv_lookup::Argument::required_degree; a blob that declares less
    would lose the top of the quotient (d - 1 pieces), so zk_pk_create recomputes it and refuses."""
    import struct
    from plonk_fixtures import build_multi_lookup_circuit
    circ, _, _ = build_multi_lookup_circuit(6, seed=3, n_inputs=1, input_degree=2, gate_degree=3)
    assert circ.degree() == 5 and max(g.degree() for g in circ.gates) == 3
    blob = bytearray(circ.blob())
    blob[16:20] = struct.pack("<I", 4)
    with pytest.raises(zk.ZkError, match="declared degree 4 is below"):
        ctx.pk_create(srs8[circ.k], bytes(blob))


def test_hostile_blobs_are_refused(zk, ctx, srs8):
    """header counts are checked against the blob before anything is sized by them"""
    import struct
    circ, _, _ = build_circuit(6, 1, False)
    blob = circ.blob()
    for field, value in ((6, 0xFFFFFF), (5, 0x7FFFFFFF), (8, 0xFFFFFFFF), (11, 0xFFFFFFFF), (10, 0xFFFFFFF0), (9, 0x40000000)):
        bad = bytearray(blob)
        bad[4 * field:4 * field + 4] = struct.pack("<I", value)
        with pytest.raises(zk.ZkError):
            ctx.pk_create(srs8[circ.k], bytes(bad))
    for cut in (40, 200, len(circ.cs_blob()) - 3, len(blob) - 32):
        with pytest.raises(zk.ZkError):
            ctx.pk_create(srs8[circ.k], blob[:cut])
    bad = bytearray(blob)
    bad[4:8] = struct.pack("<I", 2)                               # an older blob version
    with pytest.raises(zk.ZkError, match="version"):
        ctx.pk_create(srs8[circ.k], bytes(bad))


def test_caller_supplied_transcript_repr(ctx, cref, srs8, s_g2):
    """halo2 absorbs vk.transcript_repr() first [REF zkevm-circuits/src/super_circuit/test.rs:70-85];
    the Rust side computes it and installs it with zk_pk_set_transcript_repr.  With the value the
    reference pins for the SuperCircuit as the stand-in: the proof is the oracle prover's proof for
    that representative and verifies only under it."""
    from oracle import plonk_prover as pp
    circ, adv, inst = build_circuit(6, seed=9, wide=False)
    pinned = 0x1b3d158be8148c9e8ac9fce6eff2c576027c356ee1ff68ad7662d61556d5a7d7
    seed = bytes(16)
    pk = ctx.pk_create(srs8[circ.k], circ.blob())
    try:
        com, rep0 = pk.vk(circ.F + len(circ.perm_cols))
        default = _session_proof(ctx, pk, adv, inst, seed, "shplonk")
        pk.set_transcript_repr(cref.to_mont([pinned])[0])
        _, rep1 = pk.vk(circ.F + len(circ.perm_cols))
        proof = _session_proof(ctx, pk, adv, inst, seed, "shplonk")
    finally:
        pk.destroy()
    vk_points = cref.affine_from_mont(com)
    assert cref.from_mont(rep1.reshape(1, 4))[0] == pinned and cref.from_mont(rep0.reshape(1, 4))[0] == pv.default_vk_repr(circ, vk_points)
    assert proof != default
    assert proof == pp.create_proof(circ, pp.Srs(circ.k, S_SECRET), adv, inst, pinned, seed, "shplonk")
    assert pv.verify(circ, vk_points, pinned, inst, proof, s_g2, multiopen="shplonk")
    try:
        assert not pv.verify(circ, vk_points, pinned, inst, default, s_g2, multiopen="shplonk")
    except AssertionError:
        pass


@pytest.mark.parametrize("kind,name", [(1, "poseidon"), (2, "evm")])
def test_poseidon_and_evm_transcripts_equal_the_oracle_prover(ctx, cref, srs8, s_g2, kind, name):
    """gen_snark_shplonk (Poseidon transcript) and gen_evm_proof_shplonk (Keccak transcript, 64-byte
    big-endian points) [REF prover/src/common/prover/utils.rs:31], [REF prover/src/common/prover/evm.rs:67]:
    the session with the built-in transcript produces the oracle prover's bytes, and the oracle
    verifier accepts them with the same transcript only."""
    from oracle import plonk_prover as pp
    circ, adv, inst = build_circuit(6, seed=13, wide=True)
    short = [inst[0][:8]]
    seed = bytes(range(16))
    pk = ctx.pk_create(srs8[circ.k], circ.blob())
    try:
        com, rep = pk.vk(circ.F + len(circ.perm_cols))
        proof = _session_proof(ctx, pk, adv, short, seed, "shplonk", transcript_kind=kind, slices=True)
        blake = _session_proof(ctx, pk, adv, short, seed, "shplonk", slices=True)
    finally:
        pk.destroy()
    vk_points, vk_repr = cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0]
    want = pp.create_proof(circ, pp.Srs(circ.k, S_SECRET), adv, short, vk_repr, seed, "shplonk", transcript=name)
    assert proof == want and proof != blake
    if kind == 2:
        assert len(proof) > len(blake)            # uncompressed points
    assert pv.verify(circ, vk_points, vk_repr, short, proof, s_g2, multiopen="shplonk", transcript=name)


def test_degree_three_circuit(ctx, cref, srs8, s_g2):
    """cs.degree() = 3 (only the permutation argument sets it): one column per permutation chunk,
    two quotient pieces, extended domain 2n"""
    import random
    from oracle import plonk_prover as pp
    k = 5
    circ = plonk.Circuit(k, num_fixed=1, num_advice=3, num_instance=0, blinding_factors=5)
    q, a, b_, c_ = circ.fixed_col(0), circ.advice_col(0), circ.advice_col(1), circ.advice_col(2)
    circ.add_gate(q * (a + b_ - c_))
    rng = random.Random(1)
    adv = [[0] * circ.n for _ in range(3)]
    for row in range(circ.u):
        circ.fixed[0][row] = row % 2
        adv[0][row], adv[1][row] = rng.randrange(b.R_MOD), rng.randrange(b.R_MOD)
        adv[2][row] = (adv[0][row] + adv[1][row]) % b.R_MOD if row % 2 else rng.randrange(b.R_MOD)
    adv[0][4] = adv[2][3]
    adv[2][5] = (adv[0][5] + adv[1][5]) % b.R_MOD
    circ.copy((plonk.ADVICE, 2, 3), (plonk.ADVICE, 0, 4))
    circ.enable_equality(plonk.ADVICE, 1)
    assert circ.degree() == 3 and pv.check_witness(circ, adv, []) is None
    pk = ctx.pk_create(srs8[k], circ.blob())
    try:
        com, rep = pk.vk(circ.F + len(circ.perm_cols))
        assert pk.shape()["C"] == 3
        proof = _session_proof(ctx, pk, adv, [], bytes(16), "shplonk")
    finally:
        pk.destroy()
    vk_points, vk_repr = cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0]
    assert proof == pp.create_proof(circ, pp.Srs(k, S_SECRET), adv, [], vk_repr, bytes(16), "shplonk")
    assert pv.verify(circ, vk_points, vk_repr, [], proof, s_g2, multiopen="shplonk")


def test_degree_classes_leave_the_proof_unchanged(ctx, cref, srs8):
    """The quotient is evaluated class by class (constraints of degree g only on the (g - 1) n points they
    need); h, and with it every proof byte, must equal the single-domain evaluation that halo2's
    evaluate_h performs (ZK_QUOTIENT_SPLIT=0) -- on a circuit whose constraints spread over all classes
    (gates of degree 2, 3, 5 and 9, a lookup, a chunked permutation)."""
    import random
    k = 6
    circ = plonk.Circuit(k, num_fixed=3, num_advice=6, num_instance=0, blinding_factors=5)
    q, t_a, one_hot = circ.fixed_col(0), circ.fixed_col(1), circ.fixed_col(2)
    a, b_, c_, d_, e_, f_ = (circ.advice_col(i) for i in range(6))
    circ.add_gate(q * (a + b_.rot(1) - d_))                      # degree 2: a single coset
    circ.add_gate(q * (a * b_ - c_))                             # degree 3
    circ.add_gate(q * (a * b_ - c_) * (a + 1) * (b_ + 2))        # degree 5
    hi = q * (e_ * e_ - f_)
    for i in range(6):
        hi = hi * (e_ + (i + 1))
    circ.add_gate(hi)                                            # degree 9: the full extended domain
    circ.add_lookup([one_hot * a], [t_a])
    for col in range(6):
        circ.enable_equality(plonk.ADVICE, col)
    rng = random.Random(5)
    n, u = circ.n, circ.u
    adv = [[0] * n for _ in range(6)]
    for row in range(u):
        circ.fixed[1][row] = row % 16
    for row in range(u):
        circ.fixed[0][row] = 1 if row + 1 < u else 0
        circ.fixed[2][row] = row % 2
    for row in range(u):
        adv[0][row] = rng.randrange(16)
        adv[1][row] = rng.randrange(b.R_MOD)
        adv[4][row] = rng.randrange(b.R_MOD)
        adv[5][row] = adv[4][row] * adv[4][row] % b.R_MOD
    for row in range(u):
        adv[2][row] = adv[0][row] * adv[1][row] % b.R_MOD
        adv[3][row] = (adv[0][row] + adv[1][(row + 1) % n]) % b.R_MOD if circ.fixed[0][row] else 0
    adv[5][7] = adv[5][3] = adv[4][3] * adv[4][3] % b.R_MOD
    adv[4][7] = adv[4][3]
    circ.copy((plonk.ADVICE, 5, 3), (plonk.ADVICE, 5, 7))
    assert circ.degree() == 9 and pv.check_witness(circ, adv, []) is None
    from oracle import plonk_prover as pp
    proofs = {}
    # "1": classes + the additive split of sums over the classes of their terms (q (a b - c): q c is evaluated on one coset and c is
    # transformed to one coset only -- the factored form); "1-whole": classes, every constraint whole; "0": a single domain
    for mode in ("1", "1-whole", "0"):
        os.environ["ZK_QUOTIENT_SPLIT"] = mode[0]
        if mode == "1-whole":
            os.environ["ZK_QUOTIENT_ADDSPLIT"] = "0"
        try:
            pk = ctx.pk_create(srs8[k], circ.blob())
            _, rep = pk.vk(circ.F + len(circ.perm_cols))
            proofs[mode] = _session_proof(ctx, pk, adv, [], bytes(range(16)), "shplonk")
            pk.destroy()
        finally:
            os.environ.pop("ZK_QUOTIENT_SPLIT", None)
            os.environ.pop("ZK_QUOTIENT_ADDSPLIT", None)
    assert proofs["1"] == proofs["0"] and proofs["1-whole"] == proofs["0"]
    want = pp.create_proof(circ, pp.Srs(k, S_SECRET), adv, [], cref.from_mont(rep.reshape(1, 4))[0], bytes(range(16)), "shplonk")
    assert proofs["1"] == want


def test_coset_cache_running_out_of_memory_falls_back_uncached(ctx, cref, srs8):
    """the per-key coset cache must not fail a proof when the device has no room for another slot --
    it freezes (slots that exist stay in use) and the remaining key columns are transformed per proof.  The
    shortage is injected after 3 slots (ZK_PK_COSET_CACHE_FAIL_AFTER); two proofs on the same key, both equal
    to the proof of an unconstrained key, and the shared budget is returned when the key goes."""
    circ, adv, inst = build_circuit(7, seed=21, wide=True)
    seed = bytes(range(16))
    pk = ctx.pk_create(srs8[circ.k], circ.blob())
    want = _session_proof(ctx, pk, adv, inst, seed, "shplonk")
    assert _session_proof(ctx, pk, adv, inst, seed, "shplonk") == want          # second proof: every slot cached
    pk.destroy()
    os.environ["ZK_PK_COSET_CACHE_FAIL_AFTER"] = "3"
    try:
        pk = ctx.pk_create(srs8[circ.k], circ.blob())
        assert _session_proof(ctx, pk, adv, inst, seed, "shplonk") == want
        assert _session_proof(ctx, pk, adv, inst, seed, "shplonk") == want
        pk.destroy()
    finally:
        os.environ.pop("ZK_PK_COSET_CACHE_FAIL_AFTER", None)


def test_cosets_computed_ahead_leave_the_proof_unchanged(ctx, cref, srs8):
    """Round 3: cosets of the advice columns are computed during the advice phases (ZK_ADVICE_COSET_GB) and, optionally, beside
    the lookup / permutation stages (ZK_ADVICE_COSET_LATE_GB) -- scheduling only: with everything off, with the defaults and with
    both on the proof bytes are the same, and they are the oracle prover's."""
    from oracle import plonk_prover as pp
    circ, adv, inst = build_circuit(8, seed=31, wide=True)
    seed = bytes(range(16))
    proofs = {}
    for tag, env in (("off", {"ZK_ADVICE_COSET_GB": "0"}), ("default", {}), ("both", {"ZK_ADVICE_COSET_GB": "64", "ZK_ADVICE_COSET_LATE_GB": "8"}),
                     ("late only", {"ZK_ADVICE_COSET_GB": "0", "ZK_ADVICE_COSET_LATE_GB": "8"})):
        os.environ.update(env)
        try:
            pk = ctx.pk_create(srs8[circ.k], circ.blob())
            _, rep = pk.vk(circ.F + len(circ.perm_cols))
            proofs[tag] = [_session_proof(ctx, pk, adv, inst, seed, "shplonk") for _ in range(2)]      # second proof: the key's cosets are cached
            pk.destroy()
        finally:
            for name in env:
                os.environ.pop(name, None)
    want = pp.create_proof(circ, pp.Srs(circ.k, S_SECRET), adv, inst, cref.from_mont(rep.reshape(1, 4))[0], seed, "shplonk")
    for tag, (first, second) in proofs.items():
        assert first == want and second == want, tag
