"""TEST / MEASUREMENT INFRASTRUCTURE (oracle): halo2's `create_proof` restated over ARRAYS with the C primitives of
oracle/c/oracle.c (best_multiexp, best_fft, batch inversion, Kate division, a row-parallel expression evaluator; OpenMP over
the host cores) -- the CPU prover the reference runs, restated, at sizes where it can be TIMED (`bench.py` -> `proof.*.
cpu_baseline`; SURVEY 8d last row: the reference's `[Proof generation]` timer [REF circuit-benchmarks/src/super_circuit.rs:
115-134] cannot run here, no Rust).

Same protocol, same transcript, same draws as `oracle/plonk_prover.py` (the big-int restatement the GPU session is byte-equal
to): `tests/test_oracle_cpu_prover.py` requires identical proof bytes.  Structure follows upstream's CPU algorithm: the
quotient is evaluated over the whole extended domain (every column extended with one size-2^(k+e) transform, `evaluate_h`
row-parallel), h comes back with one inverse transform -- no degree classes, no coset caching: those are zkmi355's, not
halo2's.  Only tests, bench.py's cpu_baseline leg and tools may import this module."""
from __future__ import annotations

import ctypes
from typing import Dict, List, Sequence

import numpy as np

from . import bn254 as b
from . import cref
from .plonk_prover import XorShiftRng, make_transcript, newton_interpolate, product_of_differences, rotation_sets
from .plonk_verifier import ADVICE, FIXED, INSTANCE

R = b.R_MOD


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def arr(vals: Sequence[int]) -> np.ndarray:
    return cref.to_mont([v % R for v in vals])


def const(v: int) -> np.ndarray:
    return cref.fr_const(v % R)


def zeros(n: int) -> np.ndarray:
    return np.zeros((n, 4), dtype=np.uint64)


def _bin(name, a, c):
    o = np.empty_like(a)
    getattr(cref.lib(), f"orc_fe_{name}_vec_mt")(_p(a), _p(c), _p(o), ctypes.c_size_t(a.shape[0]))
    return o


def mul(a, c): return _bin("mul", a, c)
def add(a, c): return _bin("add", a, c)
def sub(a, c): return _bin("sub", a, c)


def scale_add(a, s: int, c=None):
    """a * s + c"""
    o = np.empty_like(a)
    cref.lib().orc_fe_scale_add(_p(a), _p(const(s)), _p(c) if c is not None else None, _p(o), ctypes.c_size_t(a.shape[0]))
    return o


def add_scalar(a, s: int):
    o = np.empty_like(a)
    cref.lib().orc_fe_add_scalar(_p(a), _p(const(s)), _p(o), ctypes.c_size_t(a.shape[0]))
    return o


def rotate(a, shift: int):
    o = np.empty_like(a)
    cref.lib().orc_fe_rotate(_p(a), ctypes.c_longlong(shift), _p(o), ctypes.c_size_t(a.shape[0]))
    return o


def powers(g: int, n: int, first: int = 1) -> np.ndarray:
    """first * g^i"""
    return cref.distribute_powers(np.tile(const(first), (n, 1)), g)


def random_poly_chacha(key32: bytes, n: int) -> np.ndarray:
    """bn254.fr_random_chacha(key, 0, 0, n) as a Montgomery array: the ChaCha20 blocks vectorised over the counter (numpy),
    the 512-bit -> Fr reduction per element.  Same values as the big-int restatement (tested); a CPU prover draws its blinding
    polynomial from a fast generator, so a pure-Python block function would only distort the timing."""
    import struct
    k = struct.unpack("<8I", key32)
    ctr = np.arange(n, dtype=np.uint64)
    init = [np.full(n, v, dtype=np.uint32) for v in (0x61707865, 0x3320646E, 0x79622D32, 0x6B206574, *k)]
    init += [(ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)]
    x = [v.copy() for v in init]

    def rotl(v, c):
        return (v << np.uint32(c)) | (v >> np.uint32(32 - c))

    def qr(a, b_, c, d):
        x[a] += x[b_]; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] += x[d]; x[b_] = rotl(x[b_] ^ x[c], 12)
        x[a] += x[b_]; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] += x[d]; x[b_] = rotl(x[b_] ^ x[c], 7)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    words = np.stack([a + b_ for a, b_ in zip(x, init)], axis=1)             # (n, 16) u32, little-endian 512-bit integers
    raw = words.astype("<u4").tobytes()
    vals = [int.from_bytes(raw[64 * i:64 * (i + 1)], "little") % R for i in range(n)]
    canon = np.array([[(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for v in vals], dtype=np.uint64)
    out = np.empty_like(canon)
    cref.lib().orc_fe_to_mont_vec(0, _p(canon), _p(out), ctypes.c_size_t(n))
    return out


def eval_program(prog, table: Dict[int, int], cols: List[np.ndarray], consts: np.ndarray, n: int, stride: int) -> np.ndarray:
    """prog: (op, a, b) triples of the key blob; table: column reference -> index into cols"""
    words = np.array([w for op, a, rot in prog for w in (op, table[a] if op == 1 else a, rot & 0xFFFFFFFF)], dtype=np.uint32)
    ptrs = (ctypes.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    out = np.empty((n, 4), dtype=np.uint64)
    rc = cref.lib().orc_eval_program(_p(words), ctypes.c_size_t(len(prog)), ptrs, _p(consts), ctypes.c_size_t(n), ctypes.c_longlong(stride), _p(out))
    assert rc == 0, "malformed program"
    return out


def ints(a: np.ndarray) -> List[int]:
    return cref.from_mont(a)


class Srs:
    def __init__(self, k: int, s: int):
        from .plonk_prover import Srs as _S
        base = _S(k, s)
        self.k, self.n, self.g, self.g_lagrange = k, base.n, base.g, base.g_lagrange

    def commit(self, coeffs: np.ndarray):
        return cref.affine_from_mont(cref.best_multiexp(coeffs, self.g[:coeffs.shape[0]]).reshape(1, 8))[0]

    def commit_lagrange(self, vals: np.ndarray):
        return cref.affine_from_mont(cref.best_multiexp(vals, self.g_lagrange).reshape(1, 8))[0]


def keygen(circ) -> dict:
    """What halo2's `ProvingKey` holds and `create_proof` does not recompute: fixed and sigma columns in Lagrange, coefficient
    and extended-coset form, l_0 / l_last / l_active on the extended coset, the coset's X values (keygen_pk, not timed as proving).
    `circ.fixed` may hold integer lists or (n, 4) Montgomery arrays."""
    n, k, u, d = circ.n, circ.k, circ.u, circ.degree()
    dom = b.EvaluationDomain(d, k)
    ext_k, ne = dom.extended_k, 1 << dom.extended_k

    def ext(c):
        e = zeros(ne)
        e[:n] = cref.distribute_powers(c, b.FR_ZETA)
        return cref.best_fft(e, dom.extended_omega, ext_k)
    as_arr = lambda col: np.ascontiguousarray(col) if isinstance(col, np.ndarray) else arr(col)
    fixed = [as_arr(col) for col in circ.fixed]
    sigma = [arr(col) for col in circ.sigma_columns()]
    fixed_c, sig_c = [cref.ifft(c, k) for c in fixed], [cref.ifft(c, k) for c in sigma]
    l0 = [0] * n; l0[0] = 1
    llast = [0] * n; llast[u] = 1
    lact = [1 if r < u else 0 for r in range(n)]
    return {"fixed": fixed, "sigma": sigma, "fixed_c": fixed_c, "sig_c": sig_c, "fixed_e": [ext(c) for c in fixed_c], "sig_e": [ext(c) for c in sig_c],
            "l0_e": ext(cref.ifft(arr(l0), k)), "ll_e": ext(cref.ifft(arr(llast), k)), "la_e": ext(cref.ifft(arr(lact), k)),
            "x_e": powers(dom.extended_omega, ne, b.FR_ZETA)}


def create_proof(circ, srs: Srs, advice: Sequence, instance: Sequence[Sequence[int]], vk_repr: int, seed16: bytes = bytes(16),
                 multiopen: str = "gwc", transcript: str = "blake2b", timings: dict = None, key: dict = None, vanishing: str = "one") -> bytes:
    """advice: integer lists or (n, 4) Montgomery arrays (halo2 holds the witness as field elements: converting Python integers
    is not part of proving).  key: keygen(circ), made on the fly when absent."""
    import time
    if key is None:
        key = keygen(circ)
    t_last = [time.perf_counter()]

    def mark(name):
        if timings is not None:
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + now - t_last[0]
            t_last[0] = now
    n, k, u, bf, d = circ.n, circ.k, circ.u, circ.bf, circ.degree()
    A, Pn, L = circ.A, len(circ.perm_cols), len(circ.lookups)
    chunk = d - 2
    C = (Pn + chunk - 1) // chunk if Pn else 0
    dom = b.EvaluationDomain(d, k)
    ext_k, ne = dom.extended_k, 1 << dom.extended_k
    step = ne // n
    omega = dom.omega
    gates = [circ.compile(g) for g in circ.gates]
    lookups = [([circ.compile(e) for e in lk.table], [[circ.compile(e) for e in i] for i in lk.inputs]) for lk in circ.lookups]
    sigma, fixed = key["sigma"], key["fixed"]
    rng, tr = XorShiftRng(seed16), make_transcript(transcript)

    tr.common_scalar(vk_repr)
    for col in instance:
        if u < len(col) < n:
            raise ValueError("InstanceTooLarge")
        for row in range(min(len(col), u)):
            tr.common_scalar(col[row])
    # ---- advice phases
    adv_phase = getattr(circ, "advice_phase", [0] * A)
    chal_phase = getattr(circ, "challenge_phase", [])
    challenges = [0] * len(chal_phase)
    adv = [None] * A
    for ph in range(max([0] + list(adv_phase) + list(chal_phase)) + 1):
        cols = [i for i in range(A) if adv_phase[i] == ph]
        for i in cols:
            col = advice[i]
            col = np.array(col, dtype=np.uint64, copy=True) if isinstance(col, np.ndarray) else arr(col)
            col[u:] = arr([rng.next_fr() for _ in range(u, n)])          # halo2: advice_values[n - (blinding_factors + 1)..], row u included
            adv[i] = np.ascontiguousarray(col)
        for i in cols:
            tr.write_point(srs.commit_lagrange(adv[i]))
        for ci, cp in enumerate(chal_phase):
            if cp == ph:
                challenges[ci] = tr.squeeze()
    mark("advice commits")
    inst = [arr(list(c) + [0] * (n - len(c))) for c in instance]
    # flat column table for the expression evaluator + the constants of the key (user challenges behind C_CHAL0)
    from zkevm_circuits_amd.plonk import C_CHAL0

    def tables(fx, ad, ins):
        cols, table = [], {}
        for t_, group in ((FIXED, fx), (ADVICE, ad), (INSTANCE, ins)):
            for i, col in enumerate(group):
                table[(t_ << 24) | i] = len(cols)
                cols.append(col)
        return cols, table
    lag_cols, lag_table = tables(fixed, adv, inst)
    chal_base = len(circ.consts)
    consts_arr = arr(list(circ.consts) + list(challenges)) if (circ.consts or challenges) else zeros(1)

    def fix_consts(prog):            # PUSH_CONST of a user challenge -> its slot behind the key's constants
        return [(op, chal_base + (a - C_CHAL0) if op == 2 and a >= C_CHAL0 else a, rot) for op, a, rot in prog]
    gates = [fix_consts(g) for g in gates]
    lookups = [([fix_consts(p) for p in tabs], [[fix_consts(p) for p in ins] for ins in inputs]) for tabs, inputs in lookups]

    def compress_cols(progs, cols, table, length, stride, theta):
        acc = None
        for p in progs:
            v = eval_program(p, table, cols, consts_arr, length, stride)
            acc = v if acc is None else scale_add(acc, theta, v)
        return acc

    theta = tr.squeeze()
    # ---- lookups, round 1: m
    lk_f, lk_t, lk_m = [], [], []
    for tabs, inputs in lookups:
        fs = [compress_cols(ins, lag_cols, lag_table, n, 1, theta) for ins in inputs]
        t = compress_cols(tabs, lag_cols, lag_table, n, 1, theta)
        where = {}
        for r, v in enumerate(ints(t[:u])):
            where[v] = r                          # the LAST usable row holding a value owns its multiplicity (upstream's BTreeMap)
        m = [0] * n
        for f in fs:
            for r, v in enumerate(ints(f[:u])):
                assert v in where, f"lookup input at row {r} is not in the table"
                m[where[v]] += 1
        lk_f.append(fs); lk_t.append(t); lk_m.append(arr(m))
    for m in lk_m:
        tr.write_point(srs.commit_lagrange(m))
    mark("lookup m")
    beta, gamma = tr.squeeze(), tr.squeeze()
    # ---- permutation grand products, chunked
    pz = []
    start = 1
    omega_pows = powers(omega, n) if C else None
    for c in range(C):
        num = den = None
        for j in range(c * chunk, min(Pn, (c + 1) * chunk)):
            t_, i_ = circ.perm_cols[j]
            v = lag_cols[lag_table[(t_ << 24) | i_]]
            a_ = add_scalar(scale_add(omega_pows, beta * pow(b.FR_DELTA, j, R) % R, v), gamma)
            d_ = add_scalar(scale_add(sigma[j], beta, v), gamma)
            num = a_ if num is None else mul(num, a_)
            den = d_ if den is None else mul(den, d_)
        ratio = mul(num, cref.batch_invert(den))
        z = scale_add(cref.prefix_product(ratio), start)          # z[0] = start, z[r + 1] = z[r] num[r] / den[r]
        zi = ints(z[u:u + 1])[0]
        start = zi
        if bf:
            z[n - bf:] = arr([rng.next_fr() for _ in range(bf)])
        pz.append(z)
    assert C == 0 or start == 1, "permutation argument does not close"
    for z in pz:
        tr.write_point(srs.commit_lagrange(z))
    mark("permutation")
    # ---- lookups, round 2: phi
    lk_phi = []
    for fs, t, m in zip(lk_f, lk_t, lk_m):
        stack = np.concatenate([add_scalar(f, beta) for f in fs] + [add_scalar(t, beta)])
        inv = cref.batch_invert(stack)
        g_ = None
        for a_ in range(len(fs)):
            part = inv[a_ * n:(a_ + 1) * n]
            g_ = part if g_ is None else add(g_, np.ascontiguousarray(part))
        g_ = sub(np.ascontiguousarray(g_), mul(m, np.ascontiguousarray(inv[len(fs) * n:])))
        phi = cref.prefix_sum(g_)
        assert ints(phi[u:u + 1])[0] == 0, "lookup grand sum does not close"
        if bf:
            phi[n - bf:] = arr([rng.next_fr() for _ in range(bf)])
        lk_phi.append(phi)
    for phi in lk_phi:
        tr.write_point(srs.commit_lagrange(phi))
    mark("lookup phi")
    assert vanishing in ("one", "uniform")
    if vanishing == "one":                 # the constant 1, as in the reference's own proof (plonk_prover.create_proof)
        random_coeff = zeros(n)
        random_coeff[0] = arr([1])[0]
    else:
        chacha_key = b"".join(rng.next_u32().to_bytes(4, "little") for _ in range(8))
        random_coeff = random_poly_chacha(chacha_key, n)
    tr.write_point(srs.commit(random_coeff))
    y = tr.squeeze()
    mark("random polynomial")

    # ---- coefficient forms, extended forms (EvaluationDomain::{lagrange_to_coeff, coeff_to_extended})
    to_coeff = lambda a: cref.ifft(a, k)

    def ext(c):
        e = zeros(ne)
        e[:n] = cref.distribute_powers(c, b.FR_ZETA)
        return cref.best_fft(e, dom.extended_omega, ext_k)
    fixed_c, adv_c, inst_c = key["fixed_c"], [to_coeff(c) for c in adv], [to_coeff(c) for c in inst]
    sig_c = key["sig_c"]
    pz_c, m_c, phi_c = [to_coeff(z) for z in pz], [to_coeff(m) for m in lk_m], [to_coeff(p) for p in lk_phi]
    mark("coefficient forms")
    ext_cols, ext_table = tables(key["fixed_e"], [ext(c) for c in adv_c], [ext(c) for c in inst_c])
    sig_e, pz_e, m_e, phi_e = key["sig_e"], [ext(c) for c in pz_c], [ext(c) for c in m_c], [ext(c) for c in phi_c]
    l0_e, ll_e, la_e, x_e = key["l0_e"], key["ll_e"], key["la_e"], key["x_e"]
    mark("extended forms")
    rot_last = -(bf + 1)
    # ---- evaluate_h: acc = acc * y + term, gates, permutation, lookups
    acc = zeros(ne)

    def fold(term):
        nonlocal acc
        acc = scale_add(acc, y, term)
    for g in gates:
        fold(eval_program(g, ext_table, ext_cols, consts_arr, ne, step))
    one = np.tile(const(1), (ne, 1))
    if C:
        fold(mul(l0_e, sub(one, pz_e[0])))
        zl = pz_e[C - 1]
        fold(mul(ll_e, sub(mul(zl, zl), zl)))
        for c in range(1, C):
            fold(mul(l0_e, sub(pz_e[c], rotate(pz_e[c - 1], rot_last * step))))
        for c in range(C):
            left, right = rotate(pz_e[c], step), pz_e[c]
            for jj in range(c * chunk, min(Pn, (c + 1) * chunk)):
                t_, i_ = circ.perm_cols[jj]
                v = ext_cols[ext_table[(t_ << 24) | i_]]
                left = mul(left, add_scalar(scale_add(sig_e[jj], beta, v), gamma))
                right = mul(right, add_scalar(scale_add(x_e, beta * pow(b.FR_DELTA, jj, R) % R, v), gamma))
            fold(mul(la_e, sub(left, right)))
    for l, (tabs, inputs) in enumerate(lookups):
        p0, p1, me = phi_e[l], rotate(phi_e[l], step), m_e[l]
        fi = [add_scalar(compress_cols(ins, ext_cols, ext_table, ne, step, theta), beta) for ins in inputs]
        tau = add_scalar(compress_cols(tabs, ext_cols, ext_table, ne, step, theta), beta)
        prod = fi[0]
        for f in fi[1:]:
            prod = mul(prod, f)
        sum_rest = None
        for a_ in range(len(fi)):
            term = None
            for b_ in range(len(fi)):
                if b_ != a_:
                    term = fi[b_] if term is None else mul(term, fi[b_])
            term = one if term is None else term
            sum_rest = term if sum_rest is None else add(sum_rest, term)
        lhs = mul(mul(tau, prod), sub(p1, p0))
        rhs = sub(mul(tau, sum_rest), mul(prod, me))
        fold(mul(l0_e, p0))
        fold(mul(ll_e, p0))
        fold(mul(sub(lhs, rhs), la_e))
    t_ev = arr([dom.t_evaluations[j % len(dom.t_evaluations)] for j in range(len(dom.t_evaluations))])
    h_ext = mul(acc, np.tile(t_ev, (ne // t_ev.shape[0], 1)))
    mark("evaluate_h")
    # extended_to_coeff
    hc = cref.scale(cref.best_fft(h_ext, dom.extended_omega_inv, ext_k), dom.extended_ifft_divisor)
    hc = cref.distribute_powers(hc, dom.g_coset_inv)
    assert not hc[(d - 1) * n:].any(), "quotient has more than d - 1 pieces: the circuit degree is too small"
    pieces = [np.ascontiguousarray(hc[i * n:(i + 1) * n]) for i in range(d - 1)]
    for p_ in pieces:
        tr.write_point(srs.commit(p_))
    x = tr.squeeze()
    mark("h: inverse transform + commits")

    # ---- evaluations
    point = lambda rot: x * pow(omega, rot % n, R) % R

    def ev(cf, rot, write=True):
        e = cref.eval_polynomial(cf, point(rot))
        if write:
            tr.write_scalar(e)
        return e
    adv_evals = [ev(adv_c[i], rot) for i, rot in circ.advice_queries]
    fix_evals = [ev(fixed_c[i], rot) for i, rot in circ.fixed_queries]
    random_eval = ev(random_coeff, 0)
    sigma_evals = [ev(sig_c[j], 0) for j in range(Pn)]
    z_evals = []
    for c in range(C):
        e0, e1 = ev(pz_c[c], 0), ev(pz_c[c], 1)
        z_evals.append((e0, e1, ev(pz_c[c], rot_last) if c + 1 < C else None))
    lk_evals = [(ev(phi_c[l], 0), ev(phi_c[l], 1), ev(m_c[l], 0)) for l in range(L)]
    xn = pow(x, n, R)
    hcomb = zeros(n)
    for p_ in reversed(pieces):
        hcomb = scale_add(hcomb, xn, p_)
    h_eval = ev(hcomb, 0, write=False)
    mark("evaluations")

    # polynomials are identified by position in `polys` (arrays are not hashable)
    polys, queries = [], []

    def q(cf, pt, e):
        for idx, existing in enumerate(polys):
            if existing is cf:
                break
        else:
            polys.append(cf)
            idx = len(polys) - 1
        queries.append((idx, pt, e))
    for (i, rot), e in zip(circ.advice_queries, adv_evals):
        q(adv_c[i], point(rot), e)
    for c in range(C):
        q(pz_c[c], point(0), z_evals[c][0]); q(pz_c[c], point(1), z_evals[c][1])
    for c in reversed(range(C - 1)):
        q(pz_c[c], point(rot_last), z_evals[c][2])
    for l in range(L):
        q(phi_c[l], point(0), lk_evals[l][0]); q(phi_c[l], point(1), lk_evals[l][1]); q(m_c[l], point(0), lk_evals[l][2])
    for (i, rot), e in zip(circ.fixed_queries, fix_evals):
        q(fixed_c[i], point(rot), e)
    for j in range(Pn):
        q(sig_c[j], point(0), sigma_evals[j])
    q(hcomb, point(0), h_eval); q(random_coeff, point(0), random_eval)

    def lincomb(arrs, ch):               # sum_j ch^j * arrs[j]
        acc_, pw = zeros(n), 1
        for p_ in arrs:
            acc_ = scale_add(p_, pw, acc_)
            pw = pw * ch % R
        return acc_

    def kate(cf, z):
        out = zeros(n)
        out[:n - 1] = cref.kate_division(cf, z)
        return out

    if multiopen == "gwc":
        v = tr.squeeze()
        groups = []
        for idx, pt, _ in queries:
            for g_ in groups:
                if g_[0] == pt:
                    g_[1].append(polys[idx])
                    break
            else:
                groups.append([pt, [polys[idx]]])
        for pt, ps in groups:
            tr.write_point(srs.commit(kate(lincomb(ps, v), pt)))
        mark("multiopen")
        return bytes(tr.proof)

    # ---- SHPLONK: the prover-side set construction (plonk_prover.rotation_sets) over polynomial INDICES
    yy = tr.squeeze()
    sets, super_points, eval_of = rotation_sets(queries)
    v = tr.squeeze()

    def minus_low(cf, low_coeffs):       # cf(X) - r(X) for a low-degree r
        out = cf.copy()
        if low_coeffs:
            out[:len(low_coeffs)] = sub(np.ascontiguousarray(cf[:len(low_coeffs)]), arr(low_coeffs))
        return out
    low, quot = [], []
    for points, members in sets:
        rs = [newton_interpolate(points, [eval_of(idx, p_) for p_ in points]) for idx in members]
        numer = lincomb([minus_low(polys[idx], r_) for idx, r_ in zip(members, rs)], yy)
        for z in points:
            numer = kate(numer, z)
        quot.append(numer)
        low.append(rs)
    h_x = lincomb(quot, v)
    tr.write_point(srs.commit(h_x))
    uu = tr.squeeze()
    l_x, z_diffs, vpow = zeros(n), [], 1
    for (points, members), rs in zip(sets, low):
        z_i = product_of_differences(uu, [p_ for p_ in super_points if p_ not in points])
        z_diffs.append(z_i)
        inner = lincomb([minus_low(polys[idx], [b.eval_polynomial(r_, uu)]) for idx, r_ in zip(members, rs)], yy)
        l_x = scale_add(inner, vpow * z_i % R, l_x)
        vpow = vpow * v % R
    zt_eval = product_of_differences(uu, super_points)
    l_x = scale_add(h_x, (-zt_eval) % R, l_x)
    assert cref.eval_polynomial(l_x, uu) == 0
    z0_inv = b.fr_inv(z_diffs[0])
    tr.write_point(srs.commit(cref.scale(kate(l_x, uu), z0_inv)))
    mark("multiopen")
    return bytes(tr.proof)
