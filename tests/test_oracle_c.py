"""CPU: the C restatement (oracle/c/oracle.c) against the Python big-int oracle."""
import random

import numpy as np

from oracle import bn254 as b


def test_field_ops(cref):
    rng = random.Random(1)
    for which, mod in ((0, b.R_MOD), (1, b.P_MOD)):
        xs = [rng.randrange(mod) for _ in range(300)] + [0, 1, mod - 1]
        ys = [rng.randrange(mod) for _ in range(300)] + [mod - 1, mod - 1, mod - 1]
        X, Y = cref.to_mont(xs, which), cref.to_mont(ys, which)
        assert cref.limbs_to_ints(X) == [b.to_mont(x, mod) for x in xs]
        assert cref.from_mont(cref.fe_binop("mul", which, X, Y), which) == [x * y % mod for x, y in zip(xs, ys)]
        assert cref.from_mont(cref.fe_binop("add", which, X, Y), which) == [(x + y) % mod for x, y in zip(xs, ys)]
        assert cref.from_mont(cref.fe_binop("sub", which, X, Y), which) == [(x - y) % mod for x, y in zip(xs, ys)]
        assert cref.from_mont(cref.fe_inv(which, X), which) == [pow(x, -1, mod) if x else 0 for x in xs]


def test_rand_stream_is_shared(cref):
    assert cref.limbs_to_ints(cref.rand_fr_stream(42, 64)) == b.rand_fr_stream(42, 64)


def test_fft_and_poly_helpers(cref):
    rng = random.Random(2)
    for k in (1, 3, 6, 9):
        v = [rng.randrange(b.R_MOD) for _ in range(1 << k)]
        om = b.omega_for_k(k)
        w = list(v)
        b.best_fft(w, om, k)
        assert cref.from_mont(cref.best_fft(cref.to_mont(v), om, k)) == w
        w = list(v)
        b.ifft(w, k)
        assert cref.from_mont(cref.ifft(cref.to_mont(v), k)) == w
    v = [rng.randrange(b.R_MOD) for _ in range(100)]
    x = rng.randrange(b.R_MOD)
    V = cref.to_mont(v)
    assert cref.eval_polynomial(V, x) == b.eval_polynomial(v, x)
    assert cref.from_mont(cref.kate_division(V, x)) == b.kate_division(v, x)
    v[3] = 0
    V = cref.to_mont(v)
    assert cref.from_mont(cref.batch_invert(V)) == [pow(t, -1, b.R_MOD) if t else 0 for t in v]
    pp, acc = [], 1
    for t in v:
        pp.append(acc)
        acc = acc * t % b.R_MOD
    assert cref.from_mont(cref.prefix_product(V)) == pp
    g = rng.randrange(b.R_MOD)
    assert cref.from_mont(cref.distribute_powers(V, g)) == [t * pow(g, i, b.R_MOD) % b.R_MOD for i, t in enumerate(v)]


def test_curve_and_multiexp(cref):
    rng = random.Random(3)
    n = 80
    pts = [b.g1_mul(b.G1_GEN, rng.randrange(1, b.R_MOD)) for _ in range(n)]
    pts[3] = None
    sc = [rng.randrange(b.R_MOD) for _ in range(n)]
    sc[5], sc[6], sc[7] = 0, 1, b.R_MOD - 1
    A, S = cref.affine_to_mont(pts), cref.to_mont(sc)
    assert cref.affine_from_mont(A) == pts
    assert cref.affine_from_mont(cref.g1_mul(A, S)) == [b.g1_mul(p, s) for p, s in zip(pts, sc)]
    want = b.msm_naive(sc, pts)
    for threads in (1, 3, 8):
        assert cref.affine_from_mont(cref.best_multiexp(S, A, threads)) == [want]
    assert cref.affine_from_mont(cref.srs_powers(5, 12)) == b.srs_powers(5, 12)
    J = np.concatenate([A, np.tile(cref.to_mont([1], 1), (n, 1))], axis=1)
    J[3] = 0
    dbl = cref.g1_to_affine(cref.g1_jac_double(J))
    assert cref.affine_from_mont(dbl) == [b.g1_add(p, p) for p in pts]
    add = cref.g1_to_affine(cref.g1_jac_madd(cref.g1_jac_double(J), A))
    assert cref.affine_from_mont(add) == [b.g1_mul(p, 3) for p in pts]


def test_msm_closed_form_2_14(cref):
    n, s = 1 << 14, 1234
    S, G = cref.rand_fr_stream(7, n), cref.srs_powers(s, n)
    r = cref.best_multiexp(S, G)
    acc = cref.eval_polynomial(S, s)
    assert cref.affine_from_mont(r) == [b.g1_mul(b.G1_GEN, acc)]
