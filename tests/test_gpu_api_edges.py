"""GPU: empty inputs, misuse and error reporting of the newer C-ABI entry points (every call goes
through ctypes into libzkmi355.so; errors must come back as status codes with a message, never as
a crash or a silent success)."""
import numpy as np
import pytest

from oracle import bn254

pytestmark = pytest.mark.gpu


def test_empty_batches_are_no_ops(zk, ctx, cref):
    srs = ctx.srs_setup_with_s(6, cref.fr_const(5))
    assert ctx.commit_batch(srs, [], 64).shape == (0, 8)
    assert ctx.commit_batch_h2d(srs, 1, [], [], 64).shape == (0, 8)
    assert ctx.poly_eval_batch([], 64, cref.fr_const(3)).shape == (0, 4)
    out = ctx.alloc(64)
    ctx.fr_random(bytes(32), 0, 0, out, 0)
    m = ctx.alloc(64 * 32)
    assert ctx.lookup_multiplicities(m, m, 0, m, 64) is None          # no usable rows: nothing can be missing
    assert not m.download((64, 4)).any()
    srs.destroy()


def test_fr_random_is_a_pure_function_of_key_stream_and_counter(ctx, cref):
    key = bytes(range(32))
    a, b = ctx.alloc(100 * 32), ctx.alloc(100 * 32)
    ctx.fr_random(key, 9, 1000, a, 100)
    ctx.fr_random(key, 9, 1050, b, 50)
    A, B = a.download((100, 4)), b.download((50, 4))
    assert np.array_equal(A[50:], B)                                     # counter mode: block i is independent of the launch shape
    ctx.fr_random(key, 10, 1000, b, 50)
    assert not np.array_equal(b.download((50, 4)), A[:50])               # another stream
    assert all(v < bn254.R_MOD for v in cref.from_mont(A))


def test_scatter_and_coset_argument_checks(zk, ctx, cref):
    buf = ctx.alloc(16 * 32)
    with pytest.raises(zk.ZkError, match="offset < stride"):
        ctx.fr_scatter_scaled(buf, 4, cref.fr_const(1), buf, 4, 4)
    with pytest.raises(zk.ZkError, match="two-adicity"):
        ctx.coeff_to_coset(buf, 29, cref.fr_const(1), buf)


def test_session_misuse_is_reported(zk, ctx, cref):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from plonk_fixtures import build_circuit
    from zkevm_circuits_amd import plonk, sharding
    circ, adv, inst = build_circuit(6, 1, False)
    srs = ctx.srs_setup_with_s(6, cref.fr_const(77))
    pk = ctx.pk_create(srs, circ.blob())
    adv_m = {i: plonk.column_to_mont(c) for i, c in enumerate(adv)}
    inst_m = [plonk.column_to_mont(c) for c in inst]
    try:
        sess = ctx.proof_session(pk, inst_m, bytes(16))
        with pytest.raises(zk.ZkError, match="rank < world"):
            sess.set_sharding(2, 2, sharding.ALLGATHER_FN(lambda *a: 0))
        with pytest.raises(zk.ZkError, match="unknown multi-open"):
            sess.set_multiopen(7)
        sess.advice_phase(adv_m)
        with pytest.raises(zk.ZkError, match="before the first advice phase"):
            sess.set_sharding(0, 2, sharding.ALLGATHER_FN(lambda *a: 0))
        with pytest.raises(zk.ZkError, match="already committed"):
            sess.advice_phase(adv_m)
        assert len(sess.finish()) > 500
        # a failing all-gather callback aborts the proof with a message instead of producing garbage
        sess = ctx.proof_session(pk, inst_m, bytes(16))
        bad = sharding.ALLGATHER_FN(lambda *a: 1)
        sess.set_sharding(0, 2, bad)
        with pytest.raises(zk.ZkError, match="all-gather callback failed"):
            sess.advice_phase(adv_m)
        sess.abort()
        # SRS of another size than the circuit
        other = ctx.srs_setup_with_s(7, cref.fr_const(77))
        with pytest.raises(zk.ZkError):
            ctx.pk_create(other, circ.blob())
        other.destroy()
    finally:
        pk.destroy()
        srs.destroy()


def test_pinned_host_columns_and_small_helpers(zk, ctx, cref):
    """zk_host_alloc / zk_host_free (page-locked witness columns), zk_fr_powers, zk_fr_scale."""
    n = 1 << 10
    col = ctx.host_alloc((n, 4))
    col[:] = cref.rand_fr_stream(31, n)
    srs = ctx.srs_setup_with_s(10, cref.fr_const(9))
    dev = ctx.alloc(n * 32)
    got = ctx.commit_batch_h2d(srs, 1, [col], [dev], n)[0]
    assert np.array_equal(got, ctx.commit(srs, ctx.to_device(np.array(col)), n, lagrange=True))
    assert np.array_equal(dev.download((n, 4)), np.array(col))
    ctx.host_free(col)
    # memory the caller owns, pinned in place (zk_host_register): same result, and the range can be released again
    own = np.ascontiguousarray(cref.rand_fr_stream(32, n))
    ctx.host_register(own)
    got = ctx.commit_batch_h2d(srs, 1, [own], [dev], n)[0]
    assert np.array_equal(got, ctx.commit(srs, ctx.to_device(own), n, lagrange=True))
    ctx.host_unregister(own)
    with pytest.raises(zk.ZkError):
        ctx.host_unregister(own)                       # not registered any more
    srs.destroy()
    # powers: out[i] = mul * base^i
    base, mul = 0x1234567, 0xABCDEF
    out = ctx.alloc(100 * 32)
    ctx.fr_powers(cref.fr_const(base), cref.fr_const(mul), out, 100)
    assert [int(v) for v in cref.from_mont(out.download((100, 4)))] == [mul * pow(base, i, bn254.R_MOD) % bn254.R_MOD for i in range(100)]
    ctx.fr_scale(out, cref.fr_const(7), 100)
    assert [int(v) for v in cref.from_mont(out.download((100, 4)))] == [7 * mul * pow(base, i, bn254.R_MOD) % bn254.R_MOD for i in range(100)]


def test_sharded_ntt_argument_checks_and_single_rank(zk, ctx, cref):
    """zk_ntt_sharded: world must be a power of two with n >= world^2, a failing exchange callback
    comes back as a status, and world = 1 is the ordinary transform."""
    from zkevm_circuits_amd import sharding
    k = 6
    x = cref.rand_fr_stream(77, 1 << k)
    calls = []

    def exchange(_user, send, nbytes, recv):
        calls.append(nbytes)
        return 1
    cb = sharding.ALLTOALL_FN(exchange)
    buf = ctx.to_device(x)
    for rank, world in ((0, 3), (2, 2), (0, 32)):
        with pytest.raises(zk.ZkError, match="world"):
            ctx.ntt_sharded(buf, k, rank, world, cb)
    with pytest.raises(zk.ZkError, match="world\\^2"):
        ctx.ntt_sharded(buf, 3, 0, 4, cb)
    assert not calls
    half = ctx.to_device(sharding.ntt_shard_input(x, 1, 2))
    with pytest.raises(zk.ZkError, match="callback failed"):
        ctx.ntt_sharded(half, k, 1, 2, cb)
    assert calls == [(1 << k) // 4 * 32]
    ctx.ntt_sharded(buf, k, 0, 1, cb)                    # one rank: no exchange at all
    assert len(calls) == 1
    assert np.array_equal(buf.download((1 << k, 4)), cref.best_fft(x, bn254.omega_for_k(k), k))


def test_profiling_levels_and_byte_accounting(zk, ctx, cref):
    """zk_prof_enable(2) brackets only the roofline kernels' groups; the evaluator books the bytes its launches stream"""
    k, n = 12, 1 << 12
    srs = ctx.srs_setup_with_s(k, cref.fr_const(5))
    col = ctx.to_device(cref.rand_fr_stream(3, n))
    ctx.prof_reset()
    ctx.prof_enable(2)
    ctx.commit(srs, col, n, lagrange=True)
    ctx.ntt(col, k)
    ctx.prof_enable(False)
    names = set(ctx.prof_names())
    assert names and all(nm.startswith(("ntt_", "quotient", "msm_")) for nm in names), names          # level 2: every MSM class, the NTT passes, the evaluator (round 6: the sorts and reductions too)
    assert {"msm_buckets", "msm_sort", "ntt_last"} <= names
    ctx.prof_reset()
    ctx.prof_enable(True)
    ctx.commit(srs, col, n, lagrange=True)
    out = ctx.alloc(n * 32)
    prog = np.array([(1, 0, 0), (1, 0, 1), (5, 0, 0), (9, 0, 0)], dtype=np.uint32)       # col * col(+1), folded
    ctx.quotient_eval(prog, [col.ptr], cref.fr_const(1).reshape(1, 4), k, k, out)
    ctx.prof_enable(False)
    names = set(ctx.prof_names())
    assert "msm_sort" in names and "quotient_eval" in names
    assert ctx.prof_get_bytes("quotient_eval") == (2 + 1) * n * 32          # two distinct (column, rotation) operands + the result
    assert ctx.prof_get_bytes("msm_sort") == 0
    srs.destroy()


def test_ctx_sync_joins_every_library_stream(ctx):
    """zk_ctx_sync is the one fence a Rust / C caller has: it must drain the copy, auxiliary and MSM side
    streams too, not only the main stream (a bench figure was once 3 % high because it did not)."""
    ctx.sync()
    assert ctx.streams_busy() == 0
    for role in (1, 2, 3, 4, 5):            # copy, aux, three MSM side streams: 30 ms of work each, nothing on the main stream
        ctx.debug_delay(role, 30000)
    assert ctx.streams_busy() >= 1           # still in flight (the launches above returned immediately)
    ctx.sync()
    assert ctx.streams_busy() == 0
    ctx.debug_delay(0, 2000)
    ctx.debug_delay(4, 20000)
    ctx.sync()
    assert ctx.streams_busy() == 0


def test_debug_delay_on_a_fresh_context_leaves_the_prover_usable(zk, cref):
    """zk_ctx_debug_delay may be the first thing that touches the copy / auxiliary stream of a context: the stream then comes
    with its event, and a proof on that context still works (it once recorded into a null event: every later proof failed)."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from plonk_fixtures import build_circuit
    from zkevm_circuits_amd import plonk
    fresh = zk.Context(0)
    try:
        fresh.debug_delay(1, 100)
        fresh.debug_delay(2, 100)
        circ, adv, inst = build_circuit(5, seed=3, wide=False)
        srs = fresh.srs_setup_with_s(circ.k, cref.fr_const(5))
        pk = fresh.pk_create(srs, circ.blob())
        proof = fresh.create_proof(pk, [plonk.column_to_mont(c) for c in adv], [plonk.column_to_mont(c) for c in inst])
        assert len(proof) > 500
        pk.destroy()
        srs.destroy()
    finally:
        fresh.close()
