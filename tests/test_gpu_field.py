"""GPU parity: Fr / Fq element-wise arithmetic, batch inversion and scans vs the C oracle.
Bit-exact (integer arithmetic).  Calls go through the C ABI (ctypes)."""
import random

import numpy as np
import pytest

from oracle import bn254

pytestmark = pytest.mark.gpu


def _edge_values(mod):
    return [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (mod + 1) // 2, (1 << 253), (1 << 128) - 1]


@pytest.mark.parametrize("field,mod", [(0, bn254.R_MOD), (1, bn254.P_MOD)])
def test_vec_ops_match_oracle(zk, ctx, cref, field, mod):
    rng = random.Random(1234 + field)
    n = 4096 + 37
    xs = _edge_values(mod) + [rng.randrange(mod) for _ in range(n - 9)]
    ys = list(reversed(_edge_values(mod))) + [rng.randrange(mod) for _ in range(n - 9)]
    A, B = cref.to_mont(xs, field), cref.to_mont(ys, field)
    dA, dB, dO = ctx.to_device(A), ctx.to_device(B), ctx.alloc(A.nbytes)
    for op, name in ((zk.OP_ADD, "add"), (zk.OP_SUB, "sub"), (zk.OP_MUL, "mul")):
        ctx.field_vec_op(field, op, dA, dB, dO, n)
        got = dO.download(A.shape)
        want = cref.fe_binop(name, field, A, B)
        assert np.array_equal(got, want), name


def test_large_random_mul_million(zk, ctx, cref):
    n = 1 << 20
    A, B = cref.rand_fr_stream(11, n), cref.rand_fr_stream(12, n)
    dA, dB, dO = ctx.to_device(A), ctx.to_device(B), ctx.alloc(A.nbytes)
    ctx.field_vec_op(zk.FIELD_FR, zk.OP_MUL, dA, dB, dO, n)
    assert np.array_equal(dO.download(A.shape), cref.fe_binop("mul", 0, A, B))


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 63, 64, 65, 511, 512, 513, 1000, 5000, (1 << 16) + 3, (1 << 18) - 59])
def test_batch_invert(ctx, cref, n):
    A = cref.rand_fr_stream(100 + n, n)
    if n > 4:
        A[1] = 0
        A[n - 1] = 0
    if n > 600:
        A[64:192] = 0                 # whole lanes, a whole wave's worth of one stride, hold nothing but zeros
        A[300] = cref.to_mont([1])[0]
        A[301] = cref.to_mont([bn254.R_MOD - 1])[0]
    dA = ctx.to_device(A)
    ctx.fr_batch_invert(dA, n)
    assert np.array_equal(dA.download(A.shape), cref.batch_invert(A))


@pytest.mark.parametrize("n", [1, 2, 255, 2048, 2049, 70000, 1 << 18])
def test_prefix_product_and_sum(ctx, cref, n):
    A = cref.rand_fr_stream(7 + n, n)
    dA, dZ = ctx.to_device(A), ctx.alloc(A.nbytes)
    ctx.fr_prefix_product(dA, dZ, n)
    assert np.array_equal(dZ.download(A.shape), cref.prefix_product(A))
    ctx.fr_prefix_sum(dA, dZ, n)
    assert np.array_equal(dZ.download(A.shape), cref.prefix_sum(A))


@pytest.mark.parametrize("n", [1, 2, 3, 100, 4096, 100000])
def test_eval_polynomial_and_kate_division(ctx, cref, n):
    rng = random.Random(n)
    C = cref.rand_fr_stream(900 + n, n)
    dC = ctx.to_device(C)
    for x in (rng.randrange(bn254.R_MOD), 0, 1, bn254.R_MOD - 1):
        xm = cref.fr_const(x)
        got = ctx.poly_eval(dC, n, xm)
        assert cref.from_mont(got.reshape(1, 4))[0] == cref.eval_polynomial(C, x)
        if n >= 2:
            dQ = ctx.alloc((n - 1) * 32)
            ctx.kate_division(dC, n, xm, dQ)
            assert np.array_equal(dQ.download((n - 1, 4)), cref.kate_division(C, x))


@pytest.mark.parametrize("n,count", [(1, 1), (100, 3), (512, 2), (513, 3), (4096, 7), (70001, 2), (1 << 16, 20)])
def test_eval_polynomial_batch(ctx, cref, n, count):
    """zk_poly_eval_batch == eval_polynomial applied to each column (one launch, one sync)."""
    polys = [cref.rand_fr_stream(7000 + 13 * i + n, n) for i in range(count)]
    bufs = [ctx.to_device(p) for p in polys]
    for x in (0xDEADBEEF12345 + n, 0, bn254.R_MOD - 1):
        got = ctx.poly_eval_batch(bufs, n, cref.fr_const(x))
        vals = cref.from_mont(got)
        for i in range(count):
            assert vals[i] == cref.eval_polynomial(polys[i], x), (n, i, x)


@pytest.mark.parametrize("n,count,npoints", [(1, 2, 2), (100, 5, 3), (512, 4, 4), (513, 9, 5), (70001, 6, 3), (1 << 16, 40, 15)])
def test_eval_polynomial_pairs(ctx, cref, n, count, npoints):
    """zk_poly_eval_pairs == eval_polynomial of polynomial j at point point_index[j] (all pairs in one pass; below 512 coefficients
    it goes point by point): the shape of a proof's evaluations -- many polynomials at x, a few at each rotated point, points
    0 and r - 1 among them, a point nobody uses"""
    rng = random.Random(n * 31 + count)
    polys = [cref.rand_fr_stream(8100 + 17 * i + n, n) for i in range(count)]
    bufs = [ctx.to_device(p) for p in polys]
    points = [rng.randrange(bn254.R_MOD) for _ in range(npoints)]
    points[-1] = 0
    if npoints > 2:
        points[1] = bn254.R_MOD - 1
    idx = [0 if rng.random() < 0.5 else rng.randrange(npoints) for _ in range(count)]
    idx = [i if i != npoints - 2 or npoints < 4 else 0 for i in idx]              # point npoints - 2 stays unused
    got = cref.from_mont(ctx.poly_eval_pairs(bufs, idx, cref.to_mont(points), n))
    for j in range(count):
        assert got[j] == cref.eval_polynomial(polys[j], points[idx[j]]), (n, j, idx[j])
    # the same polynomial at several points
    got = cref.from_mont(ctx.poly_eval_pairs([bufs[0]] * npoints, list(range(npoints)), cref.to_mont(points), n))
    assert [int(v) for v in got] == [cref.eval_polynomial(polys[0], x) for x in points]


@pytest.mark.parametrize("n,first", [(1, 0), (1000, 0), (4099, (1 << 32) - 7)])
def test_fr_random_chacha(ctx, cref, n, first):
    """zk_fr_random == ChaCha20 block (RFC 7539 KAT-pinned oracle) -> from_uniform_bytes."""
    key = bytes((7 * i + 3) & 0xFF for i in range(32))
    stream = 0x1122334455667788
    out = ctx.alloc(n * 32)
    ctx.fr_random(key, stream, first, out, n)
    got = cref.from_mont(out.download((n, 4)))
    idx = sorted({0, n - 1, n // 2, min(n - 1, 9)})
    for i in idx:
        assert got[i] == bn254.fr_from_uniform_bytes(bn254.chacha20_block(key, first + i, stream)), i
    if n <= 1000:
        assert [int(v) for v in got] == bn254.fr_random_chacha(key, stream, first, n)


@pytest.mark.parametrize("n,usable,distinct", [(64, 58, 10), (4096, 4090, 300), (1 << 16, (1 << 16) - 6, 50000)])
def test_lookup_multiplicities(ctx, cref, n, usable, distinct):
    """zk_lookup_multiplicities == dict-based count; a value held by several table rows credits the LAST of them
    (halo2 mv_lookup: value -> row collected into a BTreeMap, later rows overwrite); misses are reported."""
    rng = random.Random(n)
    pool = [rng.randrange(bn254.R_MOD) for _ in range(distinct)]
    table = [pool[rng.randrange(distinct)] if i >= distinct else pool[i] for i in range(n)]      # every pool value present, with repeats
    inputs = [pool[rng.randrange(distinct)] for _ in range(n)]
    first = {}
    for i in range(usable):
        first[table[i]] = i
    want = [0] * n
    for r_ in range(usable):
        want[first[inputs[r_]]] += 1
    dT, dI, dM = ctx.to_device(cref.to_mont(table)), ctx.to_device(cref.to_mont(inputs)), ctx.alloc(n * 32)
    assert ctx.lookup_multiplicities(dI, dT, usable, dM, n) is None
    assert [int(v) for v in cref.from_mont(dM.download((n, 4)))] == want
    # an input that is not in the table: the lowest offending row is reported
    bad_rows = sorted(rng.sample(range(usable), 2))
    for r_ in bad_rows:
        inputs[r_] = (max(pool) + 1 + r_) % bn254.R_MOD
        assert inputs[r_] not in first
    dI = ctx.to_device(cref.to_mont(inputs))
    assert ctx.lookup_multiplicities(dI, dT, usable, dM, n) == bad_rows[0]


@pytest.mark.parametrize("field", [0, 1])
def test_asm_products_equal_the_c_forms(zk, ctx, field):
    """The device products are one generated asm statement each (csrc/mul29_asm.hip.hpp); tests/test_host_arith.py only ever sees the C
    forms.  zk_selftest_products runs mul29 / mul29_ub / sqr29 / mul2add29 in both forms on 2^21 operand sets per seed at the
    documented lazy-reduction bounds (every sixteenth lane with every limb AT its bound), Fr and Fq: not one limb may differ."""
    import ctypes
    out = (ctypes.c_uint32 * 3)()
    total = 0
    for seed in (1, 0x9E3779B9, 0xDEADBEEF):
        ctx._ck(zk.lib().zk_selftest_products(ctx.h, ctypes.c_int(field), ctypes.c_uint32(1 << 21), ctypes.c_uint32(seed), out))
        assert (out[0], out[1]) == (0, 0), f"{out[0]} lanes differ, routines mask {out[1]:#x}"
        assert out[2] == 1 << 21
        total += out[2]
    assert total >= 10 ** 6
