"""GPU parity: NTT (halo2 best_fft / EvaluationDomain) vs the C oracle restatement, bit-exact.
Small sizes are also checked against the O(n^2) definition in the Python oracle."""
import numpy as np
import pytest

from oracle import bn254

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", list(range(0, 15)) + [16, 18, 20, 21, 22])
def test_forward_matches_best_fft(ctx, cref, k):
    n = 1 << k
    A = cref.rand_fr_stream(1000 + k, n)
    got = ctx.best_fft(A, k)
    want = cref.best_fft(A, bn254.omega_for_k(k), k) if k > 0 else A
    assert np.array_equal(got, want)


@pytest.mark.parametrize("k", [1, 2, 3, 5, 8])
def test_forward_matches_definition(ctx, cref, k):
    n = 1 << k
    A = cref.rand_fr_stream(55 + k, n)
    got = cref.from_mont(ctx.best_fft(A, k))
    assert got == bn254.ntt_naive(cref.from_mont(A), bn254.omega_for_k(k))


@pytest.mark.parametrize("k", [1, 4, 10, 11, 13, 17, 20, 21])
def test_inverse_is_lagrange_to_coeff_and_round_trips(ctx, cref, k):
    n = 1 << k
    A = cref.rand_fr_stream(2000 + k, n)
    coeff = ctx.best_fft(A, k, inverse=True)
    assert np.array_equal(coeff, cref.ifft(A, k))
    assert np.array_equal(ctx.best_fft(coeff, k), A)


def test_closed_forms_at_2_20(ctx, cref):
    """delta -> all ones; all ones -> n * delta (SURVEY 8d config 2 closed-form vectors)."""
    k, n = 20, 1 << 20
    one = cref.fr_const(1)[0]
    delta = np.zeros((n, 4), dtype=np.uint64)
    delta[0] = one
    ones = np.tile(one, (n, 1))
    assert np.array_equal(ctx.best_fft(delta, k), ones)
    want = np.zeros((n, 4), dtype=np.uint64)
    want[0] = cref.fr_const(n)[0]
    assert np.array_equal(ctx.best_fft(ones, k), want)


def test_linearity_at_2_20(zk, ctx, cref):
    k, n = 20, 1 << 20
    A, B = cref.rand_fr_stream(1, n), cref.rand_fr_stream(2, n)
    fa, fb = ctx.best_fft(A, k), ctx.best_fft(B, k)
    fab = ctx.best_fft(cref.fe_binop("add", 0, A, B), k)
    assert np.array_equal(fab, cref.fe_binop("add", 0, fa, fb))


@pytest.mark.parametrize("k,ext_k", [(3, 5), (6, 9), (10, 12), (12, 15), (16, 19)])
def test_coset_extended_domain(ctx, cref, k, ext_k):
    """EvaluationDomain::coeff_to_extended / extended_to_coeff."""
    n, ne = 1 << k, 1 << ext_k
    A = cref.rand_fr_stream(3000 + k, n)
    dA, dE = ctx.to_device(A), ctx.alloc(ne * 32)
    ctx.coeff_to_extended(dA, k, ext_k, dE)
    got = dE.download((ne, 4))
    padded = np.zeros((ne, 4), dtype=np.uint64)
    padded[:n] = cref.distribute_powers(A, bn254.FR_ZETA)
    want = cref.best_fft(padded, bn254.omega_for_k(ext_k), ext_k)
    assert np.array_equal(got, want)
    ctx.extended_to_coeff(dE, ext_k)
    back = dE.download((ne, 4))
    assert np.array_equal(back[:n], A)
    assert not back[n:].any()


def test_ntt_rejects_bad_sizes(zk, ctx):
    buf = ctx.alloc(64)
    with pytest.raises(zk.ZkError):
        ctx.ntt(buf, 29)


@pytest.mark.parametrize("k,ext_k", [(4, 6), (9, 12), (11, 12), (14, 17)])
def test_cosets_tile_the_extended_domain(ctx, cref, k, ext_k):
    """zk_coeff_to_coset (coeff_to_extended_part): coset r = zeta * omega_ext^r of H holds the
    extended-domain values at indices i * 2^(ext_k-k) + r; zk_fr_scatter_scaled interleaves them back."""
    n, ne, parts = 1 << k, 1 << ext_k, 1 << (ext_k - k)
    A = cref.rand_fr_stream(4100 + k, n)
    dA, dE, dC, dX = ctx.to_device(A), ctx.alloc(ne * 32), ctx.alloc(n * 32), ctx.alloc(ne * 32)
    ctx.coeff_to_extended(dA, k, ext_k, dE)
    want = dE.download((ne, 4))
    w_ext = bn254.omega_for_k(ext_k)
    for r in range(parts):
        g = bn254.FR_ZETA * pow(w_ext, r, bn254.R_MOD) % bn254.R_MOD
        ctx.coeff_to_coset(dA, k, cref.fr_const(g), dC)
        assert np.array_equal(dC.download((n, 4)), want[r::parts]), r
        ctx.fr_scatter_scaled(dC, n, cref.fr_const(1), dX, parts, r)
    assert np.array_equal(dX.download((ne, 4)), want)
    # a scale is applied on the way: 3 * values
    ctx.fr_scatter_scaled(dC, n, cref.fr_const(3), dX, parts, parts - 1)
    got = cref.from_mont(dX.download((ne, 4))[parts - 1::parts][:4])
    assert got == [3 * v % bn254.R_MOD for v in cref.from_mont(want[parts - 1::parts][:4])]
    # in place
    ctx.coeff_to_coset(dA, k, cref.fr_const(bn254.FR_ZETA), dA)
    assert np.array_equal(dA.download((n, 4)), want[0::parts])


def test_ntt_2_24_properties(ctx, cref):
    """Above the BASELINE size (three passes): delta -> all ones, inverse round trip on random data."""
    k, n = 24, 1 << 24
    delta = np.zeros((n, 4), dtype=np.uint64)
    delta[0] = cref.fr_const(1)[0]
    d = ctx.to_device(delta)
    ctx.ntt(d, k)
    out = d.download((n, 4))
    assert np.array_equal(out[::4097], np.repeat(cref.fr_const(1), out[::4097].shape[0], axis=0))
    A = cref.rand_fr_stream(24, n)
    d.upload(A)
    ctx.ntt(d, k)
    assert not np.array_equal(d.download((n, 4))[:64], A[:64])
    ctx.ntt(d, k, inverse=True)
    assert np.array_equal(d.download((n, 4)), A)


@pytest.mark.parametrize("k", [4, 10, 13, 16])
@pytest.mark.parametrize("pattern", ["max", "alternating", "one_hot_max", "ramp_high"])
def test_extreme_inputs(ctx, cref, k, pattern):
    """Inputs that maximise the magnitudes inside the lazily reduced butterflies (every element
    r - 1, alternating 0 / r - 1, ...): forward and inverse must still match the oracle bit for bit."""
    n, top = 1 << k, bn254.R_MOD - 1
    if pattern == "max":
        vals = [top] * n
    elif pattern == "alternating":
        vals = [top if i & 1 else 0 for i in range(n)]
    elif pattern == "one_hot_max":
        vals = [0] * n
        vals[n - 1] = top
    else:
        vals = [(top - i) % bn254.R_MOD for i in range(n)]
    A = cref.to_mont(vals)
    d = ctx.to_device(A)
    ctx.ntt(d, k)
    want = cref.best_fft(A, bn254.omega_for_k(k), k)
    assert np.array_equal(d.download((n, 4)), want)
    ctx.ntt(d, k, inverse=True)
    assert np.array_equal(d.download((n, 4)), A)


@pytest.mark.parametrize("k,count", [(0, 3), (1, 2), (5, 20), (10, 33), (13, 17), (18, 9), (20, 5), (20, 35)])
def test_batched_transforms_equal_single_ones(ctx, cref, k, count):
    """zk_ntt_batch / zk_coeff_to_coset_batch put several columns into one launch (blockIdx.y = column): every column must
    come out exactly as from the single-column entry points, across group boundaries of the batch, forward and inverse."""
    n = 1 << k
    cols = [cref.rand_fr_stream(9100 + 31 * k + i, n) for i in range(count)]
    want_f = [cref.best_fft(c, bn254.omega_for_k(k), k) for c in cols]
    bufs = [ctx.to_device(c) for c in cols]
    ctx.ntt_batch(bufs, k)
    for b, w in zip(bufs, want_f):
        assert np.array_equal(b.download((n, 4)), w)
    ctx.ntt_batch(bufs, k, inverse=True)
    for b, c in zip(bufs, cols):
        assert np.array_equal(b.download((n, 4)), c)
    g = cref.fr_const(0x5EED + k)
    outs = [ctx.alloc(n * 32) for _ in range(count)]
    ctx.coeff_to_coset_batch(bufs, k, g, outs)
    single = ctx.alloc(n * 32)
    for b, o in zip(bufs, outs):
        ctx.coeff_to_coset(b, k, g, single)
        assert np.array_equal(o.download((n, 4)), single.download((n, 4)))
    # in place as well
    ctx.coeff_to_coset_batch(bufs[:2], k, g, bufs[:2])
    assert np.array_equal(bufs[0].download((n, 4)), outs[0].download((n, 4)))
    ctx.ntt_batch([], k)


@pytest.mark.parametrize("k", list(range(7, 23)))
def test_fixed_structure_passes_equal_the_generic_kernels(ctx, cref, k):
    """csrc/ntt.hip: the passes whose step structure is fixed at compile time (k_ntt_pass_f / k_ntt_last_f, the default wherever a
    digit size has an instance) claim to be bit-identical to the run-time kernels (ZK_NTT_FIXED=0): forward, inverse and coset
    transforms of the same columns under both, a batch that crosses a launch group, and the oracle on the first column."""
    import os
    n = 1 << k
    count = 3 if k >= 20 else 5
    cols = [cref.rand_fr_stream(7700 + 13 * k + i, n) for i in range(count)]
    g = cref.fr_const(0xC05E7 + k)

    def run():
        bufs = [ctx.to_device(c) for c in cols]
        outs = [ctx.alloc(n * 32) for _ in range(count)]
        ctx.ntt_batch(bufs, k)
        fwd = [b.download((n, 4)) for b in bufs]
        ctx.ntt_batch(bufs, k, inverse=True)
        inv = [b.download((n, 4)) for b in bufs]
        ctx.coeff_to_coset_batch(bufs, k, g, outs)
        cos = [o.download((n, 4)) for o in outs]
        for b_ in bufs + outs:
            b_.free()
        return fwd, inv, cos
    default = run()
    os.environ["ZK_NTT_FIXED"] = "0"
    try:
        generic = run()
    finally:
        os.environ.pop("ZK_NTT_FIXED")
    for a_, b_ in zip(default, generic):
        for x_, y_ in zip(a_, b_):
            assert np.array_equal(x_, y_)
    assert np.array_equal(default[0][0], cref.best_fft(cols[0], bn254.omega_for_k(k), k))
    for c, v in zip(cols, default[1]):
        assert np.array_equal(c, v)
