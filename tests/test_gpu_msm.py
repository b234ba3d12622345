"""GPU parity: G1 group law, SRS generation and MSM (halo2 best_multiexp / ParamsKZG) vs the
oracle.  Bit-exact on the affine result."""
import os
import random

import numpy as np
import pytest

from oracle import bn254

pytestmark = pytest.mark.gpu

R = bn254.R_MOD


def _rand_points(cref, n, seed):
    sc = cref.rand_fr_stream(seed, n)
    gen = cref.affine_to_mont([bn254.G1_GEN] * n)
    return cref.g1_mul(gen, sc)


def test_g1_add_and_mul_vec(ctx, cref):
    n = 300
    P, Q = _rand_points(cref, n, 5), _rand_points(cref, n, 6)
    Q[0] = P[0]                                   # doubling
    Q[1] = P[1]; Q[1, 4:] = cref.to_mont([bn254.P_MOD - cref.from_mont(P[1, 4:].reshape(1, 4), 1)[0]], 1)[0]  # P + (-P)
    P[2] = 0                                      # identity + Q
    Q[3] = 0                                      # P + identity
    dP, dQ, dO = ctx.to_device(P), ctx.to_device(Q), ctx.alloc(P.nbytes)
    ctx.g1_affine_add(dP, dQ, dO, n)
    got = cref.affine_from_mont(dO.download(P.shape))
    pp, qq = cref.affine_from_mont(P), cref.affine_from_mont(Q)
    assert got == [bn254.g1_add(a, b) for a, b in zip(pp, qq)]
    assert got[1] is None
    S = cref.rand_fr_stream(77, n)
    S[0] = 0
    S[1] = cref.fr_const(1)[0]
    S[2] = cref.fr_const(R - 1)[0]
    dS = ctx.to_device(S)
    ctx.g1_mul(dQ, dS, dO, n)
    assert np.array_equal(dO.download(P.shape), cref.g1_mul(Q, S))


@pytest.mark.parametrize("k", [1, 4, 9])
def test_srs_setup_with_s(ctx, cref, k):
    s = 0x1234
    srs = ctx.srs_setup_with_s(k, cref.fr_const(s))
    n = 1 << k
    assert np.array_equal(srs.download_g(), cref.srs_powers(s, n))
    om = bn254.omega_for_k(k)
    sn = pow(s, n, R)
    lag = [pow(om, i, R) * (sn - 1) % R * bn254.fr_inv(n * (s - pow(om, i, R)) % R) % R for i in range(n)]
    gen = cref.affine_to_mont([bn254.G1_GEN] * n)
    assert np.array_equal(srs.download_g_lagrange(), cref.g1_mul(gen, cref.to_mont(lag)))
    srs.destroy()


@pytest.mark.parametrize("n", [0, 1, 2, 3, 31, 32, 33, 100, 1000, 4097])
def test_msm_small_matches_best_multiexp(ctx, cref, n):
    if n == 0:
        out = ctx.best_multiexp(np.zeros((0, 4), np.uint64), np.zeros((0, 8), np.uint64))
        assert not out.any()
        return
    S, P = cref.rand_fr_stream(40 + n, n), _rand_points(cref, n, 50 + n)
    got = ctx.best_multiexp(S, P)
    assert np.array_equal(got, cref.best_multiexp(S, P))
    if n <= 33:
        assert cref.affine_from_mont(got) == [bn254.msm_naive(cref.from_mont(S), cref.affine_from_mont(P))]


def test_msm_edge_cases(ctx, cref):
    n = 2048
    P = _rand_points(cref, n, 9)
    P[5] = 0                      # identity base
    P[7] = P[6]                   # repeated point
    P[9] = P[8]; P[9, 4:] = cref.to_mont([bn254.P_MOD - cref.from_mont(P[8, 4:].reshape(1, 4), 1)[0]], 1)[0]
    cases = {
        "zeros": np.zeros((n, 4), np.uint64),
        "ones": np.tile(cref.fr_const(1)[0], (n, 1)),
        "r_minus_1": np.tile(cref.fr_const(R - 1)[0], (n, 1)),
        "same_small": np.tile(cref.fr_const(0xFFFF)[0], (n, 1)),
    }
    rng = random.Random(3)
    witness_like = [0 if rng.random() < 0.6 else (rng.randrange(1 << 16) if rng.random() < 0.75 else rng.randrange(R)) for _ in range(n)]
    cases["witness_like"] = cref.to_mont(witness_like)
    for name, S in cases.items():
        assert np.array_equal(ctx.best_multiexp(S, P), cref.best_multiexp(S, P)), name


def test_msm_2_16_matches_oracle_pippenger(ctx, cref):
    n = 1 << 16
    S, P = cref.rand_fr_stream(1, n), cref.srs_powers(1234, n)
    assert np.array_equal(ctx.best_multiexp(S, P), cref.best_multiexp(S, P))


def test_msm_2_20_closed_form_over_srs(zk, ctx, cref):
    """BASELINE config 2 size: MSM(a, g) with g[i] = s^i G equals (sum a_i s^i) * G, and
    commit_lagrange of the evaluations of the same polynomial gives the same point."""
    k, n, s = 20, 1 << 20, 0xC0FFEE
    srs = ctx.srs_setup_with_s(k, cref.fr_const(s))
    A = cref.rand_fr_stream(2020, n)
    dA = ctx.to_device(A)
    got = ctx.commit(srs, dA, n)
    acc = cref.eval_polynomial(A, s)
    want = cref.g1_mul(cref.affine_to_mont([bn254.G1_GEN]), cref.to_mont([acc]))[0]
    assert np.array_equal(got, want)
    ctx.ntt(dA, k)                      # coefficients -> evaluations
    assert np.array_equal(ctx.commit(srs, dA, n, lagrange=True), want)
    # splitting identity: MSM over halves adds up
    h = n // 2
    lo = ctx.msm(dA.ptr, srs.g_lagrange_ptr, h)
    hi = ctx.msm(dA.ptr + 32 * h, srs.g_lagrange_ptr + 64 * h, h)
    both = bn254.g1_add(cref.affine_from_mont(lo)[0], cref.affine_from_mont(hi)[0])
    assert both == cref.affine_from_mont(want)[0]
    srs.destroy()


@pytest.mark.parametrize("kind", ["boolean", "all_equal", "small_values", "two_values"])
def test_msm_skewed_scalars_2_16(ctx, cref, kind):
    """Selector / boolean / small-value columns put a large share of the points into one bucket:
    the task split must keep the result exact (and the run time flat)."""
    import time
    n = 1 << 16
    rng = random.Random(17)
    P = cref.srs_powers(99, n)
    if kind == "boolean":
        vals = [rng.randrange(2) for _ in range(n)]
    elif kind == "all_equal":
        vals = [0x1234567] * n
    elif kind == "small_values":
        vals = [rng.randrange(256) for _ in range(n)]
    else:
        vals = [rng.choice([R - 1, 5]) for _ in range(n)]
    S = cref.to_mont(vals)
    t0 = time.perf_counter()
    got = ctx.best_multiexp(S, P)
    dt = time.perf_counter() - t0
    assert np.array_equal(got, cref.best_multiexp(S, P)), kind
    assert dt < 0.5, f"{kind}: MSM took {dt:.3f}s -- a single lane is walking a giant bucket"


def test_commit_batch_matches_single_commits(ctx, cref):
    k, n = 12, 1 << 12
    srs = ctx.srs_setup_with_s(k, cref.fr_const(4242))
    cols = [cref.rand_fr_stream(300 + i, n) for i in range(5)]
    cols[2][:] = 0                                  # an all-zero column commits to the identity
    cols[3] = cref.to_mont([1] * n)                 # a selector-like column
    bufs = [ctx.to_device(c) for c in cols]
    for lagrange in (False, True):
        single = np.stack([ctx.commit(srs, b_, n, lagrange=lagrange) for b_ in bufs])
        batch = ctx.commit_batch(srs, [b_.ptr for b_ in bufs], n, lagrange=lagrange)
        assert np.array_equal(single, batch)
        assert not batch[2].any()
    basis = srs.download_g()
    assert np.array_equal(ctx.commit_batch(srs, [bufs[0].ptr], n)[0], cref.best_multiexp(cols[0], basis))
    assert ctx.commit_batch(srs, [], n).shape == (0, 8)
    srs.destroy()


def test_commit_batch_h2d_uploads_and_commits(ctx, cref):
    """zk_commit_batch_h2d: host columns are uploaded on the copy stream under the previous MSM."""
    k, n = 13, 1 << 13
    srs = ctx.srs_setup_with_s(k, cref.fr_const(99))
    cols = [cref.rand_fr_stream(500 + i, n) for i in range(6)]
    cols[1] = cref.to_mont([0, 1] * (n // 2))
    dev = [ctx.alloc(n * 32) for _ in cols]
    for basis in (0, 1):
        got = ctx.commit_batch_h2d(srs, basis, cols, dev, n)
        want = np.stack([ctx.commit(srs, ctx.to_device(c), n, lagrange=bool(basis)) for c in cols])
        assert np.array_equal(got, want)
        for c, d in zip(cols, dev):
            assert np.array_equal(d.download(c.shape), c)
    basis_pts = srs.download_g()
    assert np.array_equal(ctx.commit_batch_h2d(srs, 0, cols[:1], dev[:1], n)[0], cref.best_multiexp(cols[0], basis_pts))
    srs.destroy()


def test_msm_2_23_closed_form_over_srs(ctx, cref):
    """Above the BASELINE size (8 GiB of window tables): commit over g[i] = s^i G equals f(s) G in
    both bases, single and batched."""
    k, n, s = 23, 1 << 23, 0xABCDEF
    srs = ctx.srs_setup_with_s(k, cref.fr_const(s))
    A = cref.rand_fr_stream(2323, n)
    dA = ctx.to_device(A)
    acc = cref.eval_polynomial(A, s)
    want = cref.g1_mul(cref.affine_to_mont([bn254.G1_GEN]), cref.to_mont([acc]))[0]
    assert np.array_equal(ctx.commit(srs, dA, n), want)
    both = ctx.commit_batch(srs, [dA.ptr, dA.ptr], n)
    assert np.array_equal(both[0], want) and np.array_equal(both[1], want)
    ctx.ntt(dA, k)
    assert np.array_equal(ctx.commit(srs, dA, n, lagrange=True), want)
    srs.destroy()


@pytest.mark.parametrize("k,small", [(1, 0), (6, 3), (11, 8)])
def test_g_to_lagrange_and_downsize(ctx, cref, k, small):
    """g_to_lagrange (inverse FFT over G1) against the closed form of unsafe_setup_with_s, whose
    Lagrange basis is L_i(s) G with a known s; ParamsKZG::downsize against a fresh setup at the
    smaller size; commit(coefficients) == commit_lagrange(evaluations) over the derived basis."""
    s = cref.fr_const(0xD0C0FFEE)
    full = ctx.srs_setup_with_s(k, s)
    g, lag = full.download_g(), full.download_g_lagrange()
    derived = ctx.srs_create(k, g)                       # no Lagrange basis supplied: derived on the device
    assert np.array_equal(derived.download_g_lagrange(), lag)
    shrunk, ref_small = full.downsize(small), ctx.srs_setup_with_s(small, s)
    assert shrunk.k == small
    assert np.array_equal(shrunk.download_g(), ref_small.download_g())
    assert np.array_equal(shrunk.download_g_lagrange(), ref_small.download_g_lagrange())
    n = 1 << small
    A = cref.rand_fr_stream(77 + k, n)
    dA = ctx.to_device(A)
    c0 = ctx.commit(shrunk, dA, n)
    ctx.ntt(dA, small)
    assert np.array_equal(ctx.commit(shrunk, dA, n, lagrange=True), c0)
    for s_ in (full, derived, shrunk, ref_small):
        s_.destroy()


def test_g_to_lagrange_2_16_timing(ctx, cref):
    k = 16
    full = ctx.srs_setup_with_s(k, cref.fr_const(12345))
    import time
    t0 = time.perf_counter()
    shrunk = full.downsize(k - 1)
    dt = time.perf_counter() - t0
    ref = ctx.srs_setup_with_s(k - 1, cref.fr_const(12345))
    assert np.array_equal(shrunk.download_g_lagrange(), ref.download_g_lagrange())
    print(f"g_to_lagrange 2^{k - 1}: {dt * 1e3:.1f} ms")
    for s_ in (full, shrunk, ref):
        s_.destroy()


@pytest.mark.parametrize("k", [10, 12, 13, 14, 16, 17, 19])      # 12, 13: the grouped small-value sort at its narrowest partitions (one / two buckets each); 19: window size 19 -- one-launch partition sort + LDS-staged scatter; 17, 19: dense columns take whole-bucket tasks (17 sits on the threshold)
def test_commit_paths_agree_with_the_oracle(ctx, cref, k):
    """Commitments over an SRS take the merged-window path (one bucket set for all windows, over the
    SRS's window table) or, for columns hinted as small-valued, the per-window path; both must give
    best_multiexp's point for every scalar distribution, alone and mixed inside one pipelined batch,
    over both bases, also for n < 2^k.  Hint 2 (runs of equal scalars) keeps the sliced sort."""
    n = 1 << k
    rng = random.Random(k)
    srs = ctx.srs_setup_with_s(k, cref.fr_const(0xABCDE + k))
    kinds = {
        "dense": cref.rand_fr_stream(70 + k, n),
        "small30": cref.to_mont([rng.randrange(1 << 30) for _ in range(n)]),
        "bytes": cref.to_mont([rng.randrange(256) for _ in range(n)]),
        "boolean": cref.to_mont([rng.randrange(2) for _ in range(n)]),
        "all_equal": cref.to_mont([0x1234567] * n),
        "zero": cref.to_mont([0] * n),
        "mostly_small": cref.to_mont([rng.randrange(R) if rng.random() < 0.01 else rng.randrange(1 << 16) for _ in range(n)]),
        "extremes": cref.to_mont([rng.choice([R - 1, 5, 1 << 253, (1 << 20) - 1, 1 << 19]) for _ in range(n)]),
    }
    names = list(kinds)
    bufs = [ctx.to_device(kinds[nm]) for nm in names]
    ptrs = [b_.ptr for b_ in bufs]
    for lagrange in (True, False):
        basis = srs.download_g_lagrange() if lagrange else srs.download_g()
        want = np.stack([cref.best_multiexp(kinds[nm], basis) for nm in names])
        for hint in (None, [0] * len(names), [1] * len(names), [2] * len(names), [i % 3 for i in range(len(names))], [(i + 1) % 2 for i in range(len(names))]):
            got = ctx.commit_batch(srs, ptrs, n, lagrange=lagrange, narrow=hint)
            bad = [nm for nm, g_, w_ in zip(names, got, want) if not np.array_equal(g_, w_)]
            assert not bad, f"hint {hint}, lagrange {lagrange}: {bad} differ from best_multiexp"
        single = np.stack([ctx.commit(srs, b_, n, lagrange=lagrange) for b_ in bufs])
        assert np.array_equal(single, want)
        m = n - 3                                           # a witness polynomial of the multi-open has n - 1 coefficients
        got = ctx.commit_batch(srs, ptrs[:2], m, lagrange=lagrange, narrow=[0, 1])
        assert np.array_equal(got, np.stack([cref.best_multiexp(kinds[nm][:m], basis[:m]) for nm in names[:2]]))
    plan = ctx.msm_plan(srs, n)
    assert plan["c"] == max(8, min(k, 22)) and plan["windows"] == (256 + plan["c"] - 1) // plan["c"]
    srs.destroy()


def test_skewed_columns_stay_fast_on_both_paths(ctx, cref):
    """selector-like and constant columns at 2^18 on the merged and the per-window path: one bucket
    (or one partition of the sort) receives everything; the run time must stay flat"""
    import time
    k, n = 18, 1 << 18
    srs = ctx.srs_setup_with_s(k, cref.fr_const(77))
    dense = ctx.to_device(cref.rand_fr_stream(5, n))
    cols = {"boolean": cref.to_mont([i & 1 for i in range(n)]), "all_equal": cref.to_mont([R - 2] * n), "bytes": cref.to_mont([i % 251 for i in range(n)])}
    basis = srs.download_g_lagrange()
    ctx.commit_batch(srs, [dense.ptr], n, lagrange=True, narrow=[0])        # tables built, clocks up
    ctx.commit_batch(srs, [dense.ptr], n, lagrange=True, narrow=[1])
    t0 = time.perf_counter()
    ctx.commit_batch(srs, [dense.ptr] * 4, n, lagrange=True, narrow=[0] * 4)
    t_dense = (time.perf_counter() - t0) / 4
    for name, col in cols.items():
        buf = ctx.to_device(col)
        want = cref.best_multiexp(col, basis)
        for hint in (0, 1):
            t0 = time.perf_counter()
            got = ctx.commit_batch(srs, [buf.ptr] * 4, n, lagrange=True, narrow=[hint] * 4)
            dt = (time.perf_counter() - t0) / 4
            assert all(np.array_equal(g_, want) for g_ in got), (name, hint)
            assert dt < 4 * t_dense + 2e-3, f"{name} on path {hint}: {dt * 1e3:.2f} ms per MSM against {t_dense * 1e3:.2f} ms for dense scalars"
    srs.destroy()


def _config2_scalar_sets(cref, n):
    """SURVEY 8(d) config 2 verbatim: (a) uniform in [0, r), (b) witness-like 60 % zero / 30 % below 2^16 / 10 % uniform,
    (c) all-ones, (d) all-(r - 1)."""
    uni = cref.rand_fr_stream(0xA11CE, n)
    rng = np.random.default_rng(0xC0FFEE)
    kind = rng.random(n)
    small = cref.to_mont([int(v) for v in rng.integers(0, 1 << 16, size=n)])
    wit = cref.rand_fr_stream(0xB0B, n)
    wit[kind < 0.9] = small[kind < 0.9]
    wit[kind < 0.6] = 0
    return {"uniform": uni, "witness_like": wit, "all_ones": np.tile(cref.fr_const(1)[0], (n, 1)), "all_r_minus_1": np.tile(cref.fr_const(R - 1)[0], (n, 1))}


def test_msm_2_20_random_bases_all_config2_scalar_sets(ctx, cref):
    """BASELINE configs[1] at full size against the oracle's best_multiexp, on bases with NO structure
    (hash-to-curve, seed 0xC0FFEE -- the closed forms over s^i G cannot see a wrong table entry or a
    mis-sorted bucket that happens to cancel): zk_msm_g1 over raw device bases, and zk_commit_batch over an
    SRS made of the same points (plan c = 20, 13 windows, scaled top window; the witness-like set also takes
    the per-window path through the hint) -- every result bit-exact."""
    k, n = 20, 1 << 20
    P = cref.hash_to_curve_points(0xC0FFEE, n)
    P2 = cref.hash_to_curve_points(0xC0FFEE + 1, n)
    sets = _config2_scalar_sets(cref, n)
    want = {name: cref.best_multiexp(S, P) for name, S in sets.items()}
    want2 = {name: cref.best_multiexp(S, P2) for name, S in sets.items()}
    srs = ctx.srs_create(k, P, P2)            # g = P, g_lagrange = P2: an SRS container over arbitrary points
    dP = ctx.to_device(P)
    dev = {name: ctx.to_device(S) for name, S in sets.items()}
    names = list(sets)
    for name in names:                        # the table-free path over raw bases
        assert np.array_equal(ctx.msm(dev[name].ptr, dP.ptr, n), want[name]), f"zk_msm_g1 {name}"
    got = ctx.commit_batch(srs, [dev[nm].ptr for nm in names], n)
    for i, nm in enumerate(names):
        assert np.array_equal(got[i], want[nm]), f"zk_commit_batch (coefficient basis) {nm}"
    got = ctx.commit_batch(srs, [dev[nm].ptr for nm in names], n, lagrange=True)
    for i, nm in enumerate(names):
        assert np.array_equal(got[i], want2[nm]), f"zk_commit_batch (Lagrange basis) {nm}"
    # hinted: the small-valued sets on the per-window path, the runs of equal scalars on the sliced sort
    hints = [0, 1, 2, 2]
    got = ctx.commit_batch(srs, [dev[nm].ptr for nm in names], n, lagrange=True, narrow=hints)
    for i, nm in enumerate(names):
        assert np.array_equal(got[i], want2[nm]), f"zk_commit_batch_hint({hints[i]}) {nm}"
    # a lone commitment (the short-latency reduction) agrees with the pipelined one
    assert np.array_equal(ctx.commit(srs, dev["uniform"], n), want["uniform"])
    for b in list(dev.values()) + [dP]:
        b.free()
    srs.destroy()


@pytest.mark.parametrize("group", ["1", "3", "8"])
def test_groups_of_small_valued_columns(ctx, cref, group):
    """Consecutive small-valued columns share one launch sequence (up to eight columns as one (8 x W)-window MSM over the
    per-window table, csrc/msm.hip group_of): runs of 1, 3, 8 + 5 and 9 hinted columns with dense columns between them, every
    group size the knob allows, n below the SRS size -- every commitment bit-exact, and identical whatever the grouping."""
    import os
    k = 13
    n = (1 << k) - 37
    rng = random.Random(99)
    srs = ctx.srs_setup_with_s(k, cref.fr_const(0x7777))
    basis = srs.download_g_lagrange()[:n]
    pattern = [1, 0, 1, 1, 1, 0] + [1] * 13 + [0, 0] + [1] * 9 + [2, 1]
    cols = []
    for i, h in enumerate(pattern):
        if h == 1:
            bits = rng.choice([1, 8, 16, 30, 64])
            cols.append(cref.to_mont([rng.randrange(1 << bits) if rng.random() < 0.7 else 0 for _ in range(n)]))
        else:
            cols.append(cref.rand_fr_stream(500 + i, n))
    want = np.stack([cref.best_multiexp(c, basis) for c in cols])
    bufs = [ctx.to_device(c) for c in cols]
    os.environ["ZK_MSM_NARROW_GROUP"] = group
    try:
        got = ctx.commit_batch(srs, [b_.ptr for b_ in bufs], n, lagrange=True, narrow=pattern)
    finally:
        os.environ.pop("ZK_MSM_NARROW_GROUP", None)
    bad = [i for i in range(len(cols)) if not np.array_equal(got[i], want[i])]
    assert not bad, f"group size {group}: columns {bad} differ from best_multiexp"
    srs.destroy()


@pytest.mark.parametrize("k", [14, 20])
def test_mixed_columns_without_hints(ctx, cref, k):
    """Witness-like columns (scalar set (b): 60 % zero / 30 % < 2^16 / 10 % uniform per cell, and the same with 0.1 %
    and 50 % field-sized cells) committed WITHOUT a hint: zk_commit_batch judges each column on the device (at most a quarter of
    sampled cells >= 2^64 -> per-window path, else merged windows); whatever it picks, the point is best_multiexp's."""
    n = 1 << k
    rng = np.random.default_rng(k)
    srs = ctx.srs_setup_with_s(k, cref.fr_const(0xFACE + k))
    basis = srs.download_g_lagrange()
    cols = []
    for frac_large in (0.001, 0.10, 0.50):
        u = rng.random(n)
        small = rng.integers(0, 1 << 16, size=n, dtype=np.uint64)
        small[u < 0.6] = 0
        col = cref.to_mont([int(v) for v in small])
        big = np.flatnonzero(u >= 1 - frac_large)
        col[big] = cref.rand_fr_stream(int(1000 * frac_large) + k, big.size)         # uniform field elements (Montgomery images of uniform values)
        cols.append(col)
    bufs = [ctx.to_device(c) for c in cols]
    want = np.stack([cref.best_multiexp(c, basis) for c in cols])
    got = ctx.commit_batch(srs, [b_.ptr for b_ in bufs], n, lagrange=True)                  # hint-free
    assert np.array_equal(got, want)
    for hint in ([1, 1, 1], [0, 0, 0]):                                                      # and both paths when forced
        assert np.array_equal(ctx.commit_batch(srs, [b_.ptr for b_ in bufs], n, lagrange=True, narrow=hint), want)
    srs.destroy()


@pytest.mark.parametrize("k", [12, 14])
def test_run_structured_columns_commit_by_their_run_ends(ctx, cref, k):
    """Columns hinted as run-structured (hint 2: permutation products) with few runs are committed by Abel summation over their
    run ends against the prefix-sum table of the basis (csrc/runs.hip); one with too many runs for that, one all zero, one a single
    run of a field-sized value over every row, and dense / small-valued neighbours in the same batch take the ordinary paths.
    Every result = best_multiexp of the same column over the same basis; both bases; and with the feature off."""
    n, s = 1 << k, 0xC0DE
    rng = np.random.default_rng(70 + k)
    srs = ctx.srs_setup_with_s(k, cref.fr_const(s))
    uni = cref.rand_fr_stream(31 + k, n)

    def runs(cuts, with_zero_run=False):
        col = np.zeros((n, 4), dtype=np.uint64)
        edges = [0] + sorted(int(c) for c in cuts) + [n]
        for j in range(len(edges) - 1):
            if edges[j] < edges[j + 1] and not (with_zero_run and j % 5 == 2):
                col[edges[j]:edges[j + 1]] = uni[j % n]
        return col
    cols = {
        "few_runs": runs(rng.choice(np.arange(1, n), size=40, replace=False)),
        "zero_runs_and_tail": runs(rng.choice(np.arange(1, n - 7), size=25, replace=False), with_zero_run=True),
        "one_run": np.repeat(uni[3:4], n, axis=0),
        "all_zero": np.zeros((n, 4), dtype=np.uint64),
        "every_row_differs": uni.copy(),                         # hinted 2 but not run-structured: n runs > n / 16
        "dense_neighbour": cref.rand_fr_stream(77, n),
        "small_neighbour": cref.to_mont([int(v) for v in rng.integers(0, 1 << 16, size=n)]),
    }
    cols["zero_runs_and_tail"][n - 6:] = uni[100:106]              # blinding rows: a few single-row runs at the end
    names = list(cols)
    hints = [2, 2, 2, 2, 2, 0, 1]
    bufs = [ctx.to_device(cols[nm]) for nm in names]
    for lagrange in (True, False):
        bases = srs.download_g_lagrange() if lagrange else srs.download_g()
        want = [cref.best_multiexp(cols[nm], bases) for nm in names]
        for env in (None, "0"):
            if env is None:
                os.environ.pop("ZK_MSM_RUNS", None)
            else:
                os.environ["ZK_MSM_RUNS"] = env
            try:
                got = ctx.commit_batch(srs, [b_.ptr for b_ in bufs], n, lagrange=lagrange, narrow=hints)
            finally:
                os.environ.pop("ZK_MSM_RUNS", None)
            bad = [nm for nm, g_, w_ in zip(names, got, want) if not np.array_equal(g_, w_)]
            assert not bad, f"lagrange {lagrange}, ZK_MSM_RUNS={env}: {bad} differ from best_multiexp"
    srs.destroy()


@pytest.mark.parametrize("k", [12, 14])
def test_columns_committed_through_their_first_differences(ctx, cref, k):
    """Columns hinted as running sums with mostly equal increments (hint 3: a lookup's phi) are committed as an MSM of
    s_j = c - (z_{j+1} - z_j) over the prefix basis plus c times a fixed point (csrc/runs.hip); a sum whose increments are all
    different, a constant column, the zero column and a column whose common increment is zero take the same entry point.
    Every result = best_multiexp of the column itself over the Lagrange basis; also with the feature off and in a mixed batch."""
    n, s = 1 << k, 0xD1FF
    rng = np.random.default_rng(90 + k)
    srs = ctx.srs_setup_with_s(k, cref.fr_const(s))
    uni = cref.from_mont(cref.rand_fr_stream(41 + k, n))          # canonical integers

    def running_sum(incs, start=0):
        out, acc = [], start
        for d in incs:
            out.append(acc)
            acc = (acc + int(d)) % R
        return out
    c = int(uni[0])
    active = set(int(i) for i in rng.choice(n - 1, size=n // 12, replace=False))
    cols = {
        "lookup_like": running_sum([int(uni[j]) if j in active else c for j in range(n)]),
        "starts_elsewhere": running_sum([int(uni[j]) if j in active else c for j in range(n)], start=int(uni[5])),
        "all_increments_differ": running_sum([int(uni[j]) for j in range(n)]),
        "constant": [int(uni[9])] * n,
        "zero": [0] * n,
        "flat_with_steps": running_sum([int(uni[j]) if j % 97 == 0 else 0 for j in range(n)]),
    }
    for nm in ("lookup_like", "starts_elsewhere"):                  # blinding rows at the end
        for t in range(1, 7):
            cols[nm][n - t] = int(uni[100 + t])
    names = list(cols)
    mont = {nm: cref.to_mont(cols[nm]) for nm in names}
    bufs = [ctx.to_device(mont[nm]) for nm in names]
    basis = srs.download_g_lagrange()
    want = [cref.best_multiexp(mont[nm], basis) for nm in names]
    for hints, env in (([3] * len(names), None), ([3] * len(names), "0"), ([3, 0, 3, 1, 2, 3], None)):
        if env is None:
            os.environ.pop("ZK_MSM_DIFF", None)
        else:
            os.environ["ZK_MSM_DIFF"] = env
        try:
            got = ctx.commit_batch(srs, [b_.ptr for b_ in bufs], n, lagrange=True, narrow=hints)
        finally:
            os.environ.pop("ZK_MSM_DIFF", None)
        bad = [nm for nm, g_, w_ in zip(names, got, want) if not np.array_equal(g_, w_)]
        assert not bad, f"hints {hints}, ZK_MSM_DIFF={env}: {bad} differ from best_multiexp"
    srs.destroy()
