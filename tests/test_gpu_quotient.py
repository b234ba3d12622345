"""GPU parity: the expression / quotient evaluator (halo2 evaluate_h restated as a postfix
program) against a big-int evaluation of the same program in the oracle."""
import random

import numpy as np
import pytest

from oracle import bn254 as b

pytestmark = pytest.mark.gpu
R = b.R_MOD


def _run_oracle(prog, cols, consts, k, ext_k, divide):
    ne, scale = 1 << ext_k, 1 << (ext_k - k)
    out = []
    tev = b.EvaluationDomain(2, k).t_evaluations if False else None
    if divide:
        zn, step = pow(b.FR_ZETA, 1 << k, R), pow(b.omega_for_k(ext_k), 1 << k, R)
        tev = [b.fr_inv((zn * pow(step, j, R) - 1) % R) for j in range(scale)]
    for i in range(ne):
        st, acc, tmp = [], 0, {}
        for op, a, bb in prog:
            if op == 1:
                rot = bb if bb < (1 << 31) else bb - (1 << 32)
                st.append(cols[a][(i + rot * scale) % ne])
            elif op == 2: st.append(consts[a])
            elif op == 3: y = st.pop(); st[-1] = (st[-1] + y) % R
            elif op == 4: y = st.pop(); st[-1] = (st[-1] - y) % R
            elif op == 5: y = st.pop(); st[-1] = st[-1] * y % R
            elif op == 6: st[-1] = (-st[-1]) % R
            elif op == 7: st[-1] = st[-1] * st[-1] % R
            elif op == 8: st[-1] = 2 * st[-1] % R
            elif op == 9: acc = (acc * consts[a] + st.pop()) % R
            elif op == 10: st[-1] = st[-1] * consts[a] % R
            elif op == 11: st[-1] = (st[-1] + consts[a]) % R
            elif op == 12: tmp[a] = st[-1]
            elif op == 13: st.append(tmp[a])
        out.append(acc * tev[i % scale] % R if divide else acc)
    return out


def _run_oracle_rows(prog, cols, consts, k, ext_k, divide, rows):
    """_run_oracle on a sample of rows (the sliced-program tests run at sizes where every row in Python would take minutes)"""
    ne, scale = 1 << ext_k, 1 << (ext_k - k)
    tev = None
    if divide:
        zn, step = pow(b.FR_ZETA, 1 << k, R), pow(b.omega_for_k(ext_k), 1 << k, R)
        tev = [b.fr_inv((zn * pow(step, j, R) - 1) % R) for j in range(scale)]
    out = []
    for i in rows:
        st, acc, tmp = [], 0, {}
        for op, a, bb in prog:
            if op == 1:
                rot = bb if bb < (1 << 31) else bb - (1 << 32)
                st.append(cols[a][(i + rot * scale) % ne])
            elif op == 2: st.append(consts[a])
            elif op == 3: y = st.pop(); st[-1] = (st[-1] + y) % R
            elif op == 4: y = st.pop(); st[-1] = (st[-1] - y) % R
            elif op == 5: y = st.pop(); st[-1] = st[-1] * y % R
            elif op == 9: acc = (acc * consts[a] + st.pop()) % R
            elif op == 10: st[-1] = st[-1] * consts[a] % R
            elif op == 12: tmp[a] = st[-1]
            elif op == 13: st.append(tmp[a])
            else: raise AssertionError(op)
        out.append(acc * tev[i % scale] % R if divide else acc)
    return out


@pytest.mark.parametrize("k,ext_k,divide", [(4, 4, False), (5, 7, True), (8, 10, True), (10, 10, False)])
def test_program_matches_oracle(zk, ctx, cref, k, ext_k, divide):
    rng = random.Random(100 * k + ext_k)
    ne, ncols = 1 << ext_k, 5
    cols = [[rng.randrange(R) for _ in range(ne)] for _ in range(ncols)]
    consts = [rng.randrange(R) for _ in range(4)] + [0, 1, R - 1]
    M32 = (1 << 32)
    prog = [
        # gate 1:  c0(rot 0) * c1(rot 1) - c2(rot -1)        -> fold with consts[0] (= "y")
        (zk.Q_PUSH_COL, 0, 0), (zk.Q_PUSH_COL, 1, 1), (zk.Q_MUL, 0, 0), (zk.Q_PUSH_COL, 2, (-1) % M32), (zk.Q_SUB, 0, 0), (zk.Q_FOLD, 0, 0),
        # gate 2:  (c3 + const1)^2 * 2 + const2 * c4(rot 3)
        (zk.Q_PUSH_COL, 3, 0), (zk.Q_ADD_CONST, 1, 0), (zk.Q_SQUARE, 0, 0), (zk.Q_DOUBLE, 0, 0),
        (zk.Q_PUSH_COL, 4, 3), (zk.Q_MUL_CONST, 2, 0), (zk.Q_ADD, 0, 0), (zk.Q_FOLD, 0, 0),
        # gate 3:  -(c0 * c0 * c0) + const3  with a deeper stack
        (zk.Q_PUSH_CONST, 3, 0), (zk.Q_PUSH_COL, 0, 0), (zk.Q_PUSH_COL, 0, 0), (zk.Q_PUSH_COL, 0, 0), (zk.Q_MUL, 0, 0), (zk.Q_MUL, 0, 0), (zk.Q_NEG, 0, 0), (zk.Q_ADD, 0, 0), (zk.Q_FOLD, 0, 0),
        # edge constants 0, 1, r-1
        (zk.Q_PUSH_CONST, 4, 0), (zk.Q_PUSH_CONST, 5, 0), (zk.Q_ADD, 0, 0), (zk.Q_PUSH_CONST, 6, 0), (zk.Q_MUL, 0, 0), (zk.Q_FOLD, 0, 0),
    ]
    dcols = [ctx.to_device(cref.to_mont(c)) for c in cols]
    out = ctx.alloc(ne * 32)
    ctx.quotient_eval(np.array(prog, dtype=np.uint32), [d.ptr for d in dcols], cref.to_mont(consts), k, ext_k, out, divide)
    got = cref.from_mont(out.download((ne, 4)))
    assert got == _run_oracle(prog, cols, consts, k, ext_k, divide)


def test_vanishing_argument_end_to_end(zk, ctx, cref):
    """A real quotient: a(X) * b(X) - c(X) vanishes on the domain, so h = (a*b - c)/(X^n - 1) is a
    polynomial: evaluate on the extended coset on the GPU, come back to coefficients, check
    a(x) b(x) - c(x) = h(x) (x^n - 1) at a random point -- the identity the verifier checks."""
    k, ext_k = 8, 9
    n, ne = 1 << k, 1 << ext_k
    A, B = cref.rand_fr_stream(1, n), cref.rand_fr_stream(2, n)
    C = cref.fe_binop("mul", 0, A, B)                       # Lagrange values satisfy the gate
    bufs = []
    for lag in (A, B, C):
        d = ctx.to_device(lag)
        ctx.ntt(d, k, inverse=True)                          # -> coefficients
        e = ctx.alloc(ne * 32)
        ctx.coeff_to_extended(d, k, ext_k, e)
        bufs.append((d, e))
    prog = [(zk.Q_PUSH_COL, 0, 0), (zk.Q_PUSH_COL, 1, 0), (zk.Q_MUL, 0, 0), (zk.Q_PUSH_COL, 2, 0), (zk.Q_SUB, 0, 0), (zk.Q_FOLD, 0, 0)]
    h = ctx.alloc(ne * 32)
    ctx.quotient_eval(np.array(prog, dtype=np.uint32), [e.ptr for _, e in bufs], cref.to_mont([1]), k, ext_k, h, True)
    ctx.extended_to_coeff(h, ext_k)
    hc = h.download((ne, 4))
    assert not hc[n:].any()                                   # deg h < n  (deg(a*b) < 2n)
    x = 0x1234567
    ev = [cref.from_mont(ctx.poly_eval(d, n, cref.fr_const(x)).reshape(1, 4))[0] for d, _ in bufs]
    hx = cref.eval_polynomial(np.ascontiguousarray(hc[:n]), x)
    assert (ev[0] * ev[1] - ev[2]) % R == hx * (pow(x, n, R) - 1) % R


def test_bad_programs_are_rejected(zk, ctx, cref):
    out = ctx.alloc(32 * 16)
    col = ctx.to_device(cref.rand_fr_stream(1, 16))
    for prog in ([(zk.Q_ADD, 0, 0)], [(zk.Q_PUSH_COL, 3, 0)], [(zk.Q_PUSH_CONST, 9, 0)], [(99, 0, 0)], [(zk.Q_FOLD, 0, 0)]):
        with pytest.raises(zk.ZkError):
            ctx.quotient_eval(np.array(prog, dtype=np.uint32), [col.ptr], cref.to_mont([1]), 4, 4, out)


@pytest.mark.parametrize("k,ext_k", [(6, 6), (7, 9)])
def test_intermediates_shared_between_gates(zk, ctx, cref, k, ext_k):
    """TEE_TMP / PUSH_TMP (halo2's GraphEvaluator intermediates): a product computed by one gate and
    parked in the row's scratch is read back by later gates; same values as recomputing it."""
    rng = random.Random(7 * k + ext_k)
    ne = 1 << ext_k
    cols = [[rng.randrange(R) for _ in range(ne)] for _ in range(3)]
    consts = [rng.randrange(R) for _ in range(3)]
    M32 = 1 << 32
    shared = [(zk.Q_PUSH_COL, 0, 0), (zk.Q_PUSH_COL, 1, 1), (zk.Q_MUL, 0, 0), (zk.Q_PUSH_COL, 2, (-2) % M32), (zk.Q_ADD, 0, 0)]      # s = c0 * c1(+1) + c2(-2)
    plain = (shared + [(zk.Q_PUSH_COL, 2, 0), (zk.Q_MUL, 0, 0), (zk.Q_FOLD, 0, 0)]                       # gate 1: s * c2
             + shared + shared + [(zk.Q_MUL, 0, 0), (zk.Q_ADD_CONST, 1, 0), (zk.Q_FOLD, 0, 0)]           # gate 2: s * s + k1
             + [(zk.Q_PUSH_COL, 1, 0)] + shared + [(zk.Q_SUB, 0, 0), (zk.Q_MUL_CONST, 2, 0), (zk.Q_FOLD, 0, 0)])   # gate 3: (c1 - s) * k2
    with_tmp = (shared + [(zk.Q_TEE_TMP, 5, 0), (zk.Q_PUSH_COL, 2, 0), (zk.Q_MUL, 0, 0), (zk.Q_FOLD, 0, 0)]
                + [(zk.Q_PUSH_TMP, 5, 0), (zk.Q_PUSH_TMP, 5, 0), (zk.Q_MUL, 0, 0), (zk.Q_TEE_TMP, 0, 0), (zk.Q_ADD_CONST, 1, 0), (zk.Q_FOLD, 0, 0)]
                + [(zk.Q_PUSH_COL, 1, 0), (zk.Q_PUSH_TMP, 5, 0), (zk.Q_SUB, 0, 0), (zk.Q_MUL_CONST, 2, 0), (zk.Q_FOLD, 0, 0)])
    dcols = [ctx.to_device(cref.to_mont(c)) for c in cols]
    out = ctx.alloc(ne * 32)
    want = _run_oracle(plain, cols, consts, k, ext_k, ext_k > k)
    for prog in (plain, with_tmp):
        ctx.quotient_eval(np.array(prog, dtype=np.uint32), [d.ptr for d in dcols], cref.to_mont(consts), k, ext_k, out, ext_k > k)
        assert cref.from_mont(out.download((ne, 4))) == want
    assert _run_oracle(with_tmp, cols, consts, k, ext_k, ext_k > k) == want
    for bad, msg in (([(zk.Q_PUSH_TMP, 0, 0), (zk.Q_FOLD, 0, 0)], "read before it is written"),
                     ([(zk.Q_TEE_TMP, 0, 0)], "bad TEE_TMP"),
                     ([(zk.Q_PUSH_COL, 0, 0), (zk.Q_TEE_TMP, 1 << 20, 0), (zk.Q_FOLD, 0, 0)], "bad TEE_TMP")):
        with pytest.raises(zk.ZkError, match=msg):
            ctx.quotient_eval(np.array(bad, dtype=np.uint32), [d.ptr for d in dcols], cref.to_mont(consts), k, ext_k, out, False)


@pytest.mark.parametrize("k,ext_k,divide,slices", [(11, 11, False, 2), (11, 12, True, 3), (11, 12, True, 64), (12, 12, False, 5)])
def test_sliced_program_matches_oracle(zk, ctx, cref, k, ext_k, divide, slices, monkeypatch):
    """Round 6: a sum of terms cut into slices (ZK_QUOTIENT_SLICES forces the count; each slice evaluated from acc = 0 by its own workgroups over the same
    rows, partial sums put together with the products of the slices' fold constants).  Parked intermediates block the cuts they are alive across and are
    renumbered per slice; fold constants differ per term."""
    rng = random.Random(7 * k + ext_k + slices)
    ne, ncols = 1 << ext_k, 6
    cols = [[rng.randrange(R) for _ in range(ne)] for _ in range(ncols)]
    consts = [rng.randrange(R) for _ in range(6)]
    M32 = 1 << 32
    prog = []
    for t in range(40):
        a, b_, c = rng.randrange(ncols), rng.randrange(ncols), rng.randrange(ncols)
        prog += [(zk.Q_PUSH_COL, a, 0), (zk.Q_PUSH_COL, b_, rng.choice([0, 1, (-1) % M32])), (zk.Q_MUL, 0, 0)]
        if t % 7 == 2:      # park a product, read it back in this term and in the next two (no cut in between)
            prog += [(zk.Q_TEE_TMP, t % 3, 0)]
        if t % 7 in (3, 4):
            prog += [(zk.Q_PUSH_TMP, (t - (t % 7 - 2)) % 3, 0), (zk.Q_ADD, 0, 0)]
        prog += [(zk.Q_PUSH_COL, c, 0), (zk.Q_SUB, 0, 0), (zk.Q_MUL_CONST, rng.randrange(6), 0), (zk.Q_FOLD, rng.randrange(6), 0)]
    monkeypatch.setenv("ZK_QUOTIENT_SLICES", str(slices))
    dcols = [ctx.to_device(cref.to_mont(c)) for c in cols]
    out = ctx.alloc(ne * 32)
    ctx.quotient_eval(np.array(prog, dtype=np.uint32), [d.ptr for d in dcols], cref.to_mont(consts), k, ext_k, out, divide)
    got = cref.from_mont(out.download((ne, 4)))
    monkeypatch.setenv("ZK_QUOTIENT_SLICES", "0")
    out2 = ctx.alloc(ne * 32)
    ctx.quotient_eval(np.array(prog, dtype=np.uint32), [d.ptr for d in dcols], cref.to_mont(consts), k, ext_k, out2, divide)
    assert got == cref.from_mont(out2.download((ne, 4)))
    sample = list(range(0, ne, max(1, ne // 64)))
    want = _run_oracle_rows(prog, cols, consts, k, ext_k, divide, sample)
    assert [got[i] for i in sample] == want
