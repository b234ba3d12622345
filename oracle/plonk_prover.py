"""CPU restatement of halo2's `create_proof` (KZG, Blake2b transcript, GWC or SHPLONK multi-open)
over Python integers -- TEST INFRASTRUCTURE ONLY: never imported by the product package.

Restates `halo2_proofs::plonk::prover::create_proof` as SURVEY Appendix B describes it (advice
commitments with blinding rows, mv-lookup / logUp m and phi, chunked permutation grand products,
random vanishing polynomial, quotient split in d - 1 pieces, evaluations, multi-open) for the
single-phase-or-more sessions of `csrc/prover.hip`, with the same transcript order and the same
draws from the same generators (rand_xorshift for the blinding rows, ChaCha20 counter mode for the
blinding polynomial), so that for a given seed the GPU session and this prover must produce the
SAME BYTES.  The verifier half is `oracle/plonk_verifier.py`; the two share the expression
evaluator and the constraint order.

Sizes: pure Python, O(n) big-int work per column -- k <= 8 in the tests.
"""
from __future__ import annotations

import hashlib
import struct
from typing import List, Sequence

from . import bn254 as b
from .plonk_verifier import ADVICE, FIXED, INSTANCE, Consts, compress, eval_program

R = b.R_MOD
Q_PUSH_COL = 1


# ------------------------------------------------------------------------------------ transcript / RNG
from .transcripts import Blake2b as Blake2bWrite, make as make_transcript  # noqa: E402,F401


# ------------------------------------------------------------------------------------ SHPLONK, prover side
# Written independently of the verifier's `shplonk_sets` / `_interpolate` / `_vanishing_at` (oracle/plonk_verifier.py), which are
# pinned by the reference-produced proof (tests/test_reference_chunk_proof.py): a convention this side gets wrong shows up as a
# proof that verifier rejects.
def rotation_sets(queries):
    """halo2 poly::kzg::multiopen::shplonk `construct_intermediate_sets`, prover side.  queries: (polynomial object, point, eval)
    in the order `create_proof` lists them.  Returns ([(points ascending, [polynomial objects])...] in order of first
    appearance, all points ascending, eval lookup)."""
    order, where = [], {}                  # polynomial identity -> its points in query order
    evals = {}
    ident = lambda o: ("i", o) if isinstance(o, int) else ("o", id(o))      # coefficient lists by identity, column indices by value
    for obj, pt, e in queries:
        key = ident(obj)
        if key not in where:
            where[key] = (obj, [])
            order.append(key)
        if pt not in where[key][1]:
            where[key][1].append(pt)
        evals.setdefault((key, pt), e)
    grouped, index = [], {}                # frozenset of points -> position
    for key in order:
        obj, pts = where[key]
        fs = frozenset(pts)
        if fs not in index:
            index[fs] = len(grouped)
            grouped.append((sorted(fs), []))
        grouped[index[fs]][1].append(obj)
    every = sorted({pt for _, pt, _ in queries})
    return grouped, every, (lambda obj, pt: evals[(ident(obj), pt)])


def newton_interpolate(xs, ys):
    """coefficients (low first) of the polynomial of degree < len(xs) through (xs[i], ys[i]): divided differences, then the
    Newton form expanded from the innermost bracket out"""
    m = len(xs)
    dd = [v % R for v in ys]
    for lvl in range(1, m):
        for i in range(m - 1, lvl - 1, -1):
            dd[i] = (dd[i] - dd[i - 1]) * b.fr_inv((xs[i] - xs[i - lvl]) % R) % R
    out = [dd[m - 1]] if m else []
    for i in range(m - 2, -1, -1):         # out = out * (X - xs[i]) + dd[i]
        nxt = [0] * (len(out) + 1)
        for t, c_ in enumerate(out):
            nxt[t + 1] = (nxt[t + 1] + c_) % R
            nxt[t] = (nxt[t] - c_ * xs[i]) % R
        nxt[0] = (nxt[0] + dd[i]) % R
        out = nxt
    return out


def product_of_differences(at, points):
    out = 1
    for z_ in points:
        out = out * ((at - z_) % R) % R
    return out


class XorShiftRng:
    """rand_xorshift 0.3 (the reference prover's `gen_rng`, prover/src/utils.rs:192-195) and
    halo2curves' Fr::random = from_uniform_bytes of eight next_u64 words."""

    def __init__(self, seed16: bytes):
        s = list(struct.unpack("<4I", seed16))
        if not any(s):
            s = [0x193A6754, 0xA8A7D469, 0x97830E05, 0x113BA7BB]
        self.x, self.y, self.z, self.w = s

    def next_u32(self) -> int:
        t = (self.x ^ (self.x << 11)) & 0xFFFFFFFF
        self.x, self.y, self.z = self.y, self.z, self.w
        self.w = (self.w ^ (self.w >> 19) ^ (t ^ (t >> 8))) & 0xFFFFFFFF
        return self.w

    def next_u64(self) -> int:
        lo = self.next_u32()
        hi = self.next_u32()
        return (hi << 32) | lo

    def next_fr(self) -> int:
        return b.fr_from_uniform_bytes(b"".join(self.next_u64().to_bytes(8, "little") for _ in range(8)))


# ------------------------------------------------------------------------------------ SRS
class Srs:
    """unsafe_setup_with_s: g[i] = s^i G and the Lagrange basis L_i(s) G (closed form, known s)."""

    def __init__(self, k: int, s: int):
        from . import cref
        self.k, self.n, self.s = k, 1 << k, s
        n, w = self.n, b.omega_for_k(k)
        self.g = cref.srs_powers(s, n)
        sn1 = (pow(s, n, R) - 1) % R
        scal, wi = [], 1
        for _ in range(n):
            den = n * (s - wi) % R
            scal.append(wi * sn1 % R * b.fr_inv(den) % R if den else 1)
            wi = wi * w % R
        gen = cref.affine_to_mont([b.G1_GEN] * n)
        self.g_lagrange = cref.g1_mul(gen, cref.to_mont(scal))

    def commit(self, coeffs: Sequence[int]):
        from . import cref
        return cref.affine_from_mont(cref.best_multiexp(cref.to_mont(list(coeffs)), self.g[:len(coeffs)]).reshape(1, 8))[0]

    def commit_lagrange(self, vals: Sequence[int]):
        from . import cref
        return cref.affine_from_mont(cref.best_multiexp(cref.to_mont(list(vals)), self.g_lagrange).reshape(1, 8))[0]


def vk_commitments(circ, srs: Srs):
    """fixed then sigma commitments, as `zk_pk_vk` returns them"""
    sig = circ.sigma_columns()
    return [srs.commit_lagrange(col) for col in circ.fixed] + [srs.commit_lagrange(col) for col in sig]


# ------------------------------------------------------------------------------------ the prover
def create_proof(circ, srs: Srs, advice: Sequence[Sequence[int]], instance: Sequence[Sequence[int]], vk_repr: int,
                 seed16: bytes = bytes(16), multiopen: str = "gwc", transcript: str = "blake2b", phase_witness=None, vanishing: str = "one") -> bytes:
    """vanishing: the vanishing argument's "random" polynomial -- "one" = the constant 1 (commitment g[0], evaluation 1: what the
    reference's own proof carries, tests/test_reference_chunk_proof.py), "uniform" = n uniform coefficients (upstream PSE halo2).
    phase_witness(phase, challenges so far) -> {advice column: values}: the columns of a later phase, synthesised once
    the challenges they depend on exist (what halo2 does by calling Circuit::synthesize once per phase)."""
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
    sigma = circ.sigma_columns()
    rng, tr = XorShiftRng(seed16), make_transcript(transcript)

    # halo2 absorbs exactly the instance values it is given (KZG: QUERY_INSTANCE = false) and pads the
    # column with zeros; more values than usable rows is Error::InstanceTooLarge.  An n-row column
    # image stands for "every usable row is a public input".
    tr.common_scalar(vk_repr)
    for col in instance:
        if u < len(col) < n:
            raise ValueError("InstanceTooLarge")
        for row in range(min(len(col), u)):
            tr.common_scalar(col[row])
    # ---- advice phases: blinding rows, commitments, then the phase's challenges
    adv = [list(c) for c in advice]
    adv_phase = getattr(circ, "advice_phase", [0] * A)
    chal_phase = getattr(circ, "challenge_phase", [])
    challenges = [0] * len(chal_phase)
    for ph in range(max([0] + list(adv_phase) + list(chal_phase)) + 1):
        cols = [i for i in range(A) if adv_phase[i] == ph]
        if phase_witness is not None:
            for i, col in phase_witness(ph, list(challenges)).items():
                assert adv_phase[i] == ph
                adv[i] = list(col)
        for i in cols:
            for row in range(u, n):          # halo2: advice_values[n - (blinding_factors + 1)..], row u included
                adv[i][row] = rng.next_fr()
        for i in cols:
            tr.write_point(srs.commit_lagrange(adv[i]))
        for ci, cp in enumerate(chal_phase):
            if cp == ph:
                challenges[ci] = tr.squeeze()
    consts = Consts(circ.consts, challenges)
    inst = [list(c) + [0] * (n - len(c)) for c in instance]
    lag_cols = {FIXED: circ.fixed, ADVICE: adv, INSTANCE: inst}

    def row_lookup(row):
        return lambda t, i, rot: lag_cols[t][i][(row + rot) % n]

    theta = tr.squeeze()
    # ---- lookups, round 1 (mv_lookup::prover::Argument::prepare): theta-compressed inputs and table,
    # multiplicities m.  A table value that occurs in several usable rows is counted at its LAST row
    # (halo2 collects value -> row into a BTreeMap, later rows overwrite); rows >= u of m stay zero.
    lk_f, lk_t, lk_m = [], [], []
    for tabs, inputs in lookups:
        fs = [[compress([eval_program(p, row_lookup(r), consts) for p in ins], theta) for r in range(n)] for ins in inputs]
        t = [compress([eval_program(p, row_lookup(r), consts) for p in tabs], theta) for r in range(n)]
        where = {}
        for r in range(u):
            where[t[r]] = r
        m = [0] * n
        for f in fs:
            for r in range(u):
                assert f[r] in where, f"lookup input at row {r} is not in the table"
                m[where[f[r]]] += 1
        lk_f.append(fs); lk_t.append(t); lk_m.append(m)
    for m in lk_m:
        tr.write_point(srs.commit_lagrange(m))
    beta, gamma = tr.squeeze(), tr.squeeze()
    # ---- permutation grand products, chunked
    pz = []
    start = 1
    for c in range(C):
        z = [1] * n
        for r in range(n - 1):
            num = den = 1
            for j in range(c * chunk, min(Pn, (c + 1) * chunk)):
                t_, i_ = circ.perm_cols[j]
                v = lag_cols[t_][i_][r]
                num = num * ((v + beta * pow(b.FR_DELTA, j, R) % R * pow(omega, r, R) + gamma) % R) % R
                den = den * ((v + beta * sigma[j][r] + gamma) % R) % R
            z[r + 1] = z[r] * num % R * b.fr_inv(den) % R
        z = [v * start % R for v in z]
        start = z[u]
        for r in range(n - bf, n):
            z[r] = rng.next_fr()
        pz.append(z)
    assert C == 0 or start == 1, "permutation argument does not close"
    for z in pz:
        tr.write_point(srs.commit_lagrange(z))
    # ---- lookups, round 2 (Prepared::commit_grand_sum): phi[0] = 0, phi[r+1] = phi[r] + sum_i 1/(f_i+beta) - m/(t+beta)
    lk_phi = []
    for fs, t, m in zip(lk_f, lk_t, lk_m):
        phi = [0] * n
        for r in range(n - 1):
            g_ = (sum(b.fr_inv((f[r] + beta) % R) for f in fs) - m[r] * b.fr_inv((t[r] + beta) % R)) % R
            phi[r + 1] = (phi[r] + g_) % R
        assert phi[u] == 0, "lookup grand sum does not close"
        for r in range(n - bf, n):
            phi[r] = rng.next_fr()
        lk_phi.append(phi)
    for phi in lk_phi:
        tr.write_point(srs.commit_lagrange(phi))
    # ---- vanishing argument: the constant 1, or a blinding polynomial from ChaCha20 in counter mode
    assert vanishing in ("one", "uniform")
    if vanishing == "one":
        random_coeff = [1] + [0] * (n - 1)
    else:
        key = b"".join(rng.next_u32().to_bytes(4, "little") for _ in range(8))
        random_coeff = b.fr_random_chacha(key, 0, 0, n)
    tr.write_point(srs.commit(random_coeff))
    y = tr.squeeze()

    # ---- quotient: numerator on the extended coset, divided by X^n - 1
    to_coeff = dom.lagrange_to_coeff
    coeff = {(t_, i): to_coeff(col) for t_, cols in lag_cols.items() for i, col in enumerate(cols)}
    sig_coeff = [to_coeff(col) for col in sigma]
    pz_coeff, m_coeff, phi_coeff = [to_coeff(z) for z in pz], [to_coeff(m) for m in lk_m], [to_coeff(p) for p in lk_phi]
    ext = lambda c: dom.coeff_to_extended(c)
    ext_cols = {key_: ext(c) for key_, c in coeff.items()}
    sig_ext, pz_ext, m_ext, phi_ext = [ext(c) for c in sig_coeff], [ext(c) for c in pz_coeff], [ext(c) for c in m_coeff], [ext(c) for c in phi_coeff]
    l0 = [0] * n; l0[0] = 1
    llast = [0] * n; llast[u] = 1
    lact = [1 if r < u else 0 for r in range(n)]
    l0_e, ll_e, la_e = ext(to_coeff(l0)), ext(to_coeff(llast)), ext(to_coeff(lact))
    x_e = [b.FR_ZETA * pow(dom.extended_omega, j, R) % R for j in range(ne)]
    rot_last = -(bf + 1)
    h_ext = [0] * ne
    for j in range(ne):
        at = lambda vec, rot=0: vec[(j + rot * step) % ne]
        col_at = lambda t_, i, rot: ext_cols[(t_, i)][(j + rot * step) % ne]
        acc = 0
        for g in gates:
            acc = (acc * y + eval_program(g, col_at, consts)) % R
        if C:
            acc = (acc * y + l0_e[j] * (1 - at(pz_ext[0]))) % R
            zl = at(pz_ext[C - 1])
            acc = (acc * y + ll_e[j] * (zl * zl - zl)) % R
            for c in range(1, C):
                acc = (acc * y + l0_e[j] * (at(pz_ext[c]) - at(pz_ext[c - 1], rot_last))) % R
            for c in range(C):
                left, right = at(pz_ext[c], 1), at(pz_ext[c])
                for jj in range(c * chunk, min(Pn, (c + 1) * chunk)):
                    t_, i_ = circ.perm_cols[jj]
                    v = col_at(t_, i_, 0)
                    left = left * ((v + beta * sig_ext[jj][j] + gamma) % R) % R
                    right = right * ((v + beta * pow(b.FR_DELTA, jj, R) % R * x_e[j] + gamma) % R) % R
                acc = (acc * y + la_e[j] * (left - right)) % R
        for l, (tabs, inputs) in enumerate(lookups):
            # evaluation.rs: lhs = tau prod(phi_i) (phi(wX) - phi(X)),  rhs = prod(phi_i) (tau sum 1/phi_i - m)
            p0, p1, me = at(phi_ext[l]), at(phi_ext[l], 1), at(m_ext[l])
            fi = [(compress([eval_program(p, col_at, consts) for p in ins], theta) + beta) % R for ins in inputs]
            tau = (compress([eval_program(p, col_at, consts) for p in tabs], theta) + beta) % R
            prod = 1
            for f in fi:
                prod = prod * f % R
            sum_rest = 0                       # prod(phi_i) * sum 1/phi_i as a polynomial expression
            for a_ in range(len(fi)):
                term = 1
                for b_ in range(len(fi)):
                    if b_ != a_:
                        term = term * fi[b_] % R
                sum_rest = (sum_rest + term) % R
            lhs = tau * prod % R * (p1 - p0) % R
            rhs = (tau * sum_rest - prod * me) % R
            acc = (acc * y + l0_e[j] * p0) % R
            acc = (acc * y + ll_e[j] * p0) % R
            acc = (acc * y + (lhs - rhs) * la_e[j]) % R
        h_ext[j] = acc * dom.t_evaluations[j % len(dom.t_evaluations)] % R
    h_coeff = dom.extended_to_coeff(h_ext)
    assert not any(h_coeff[(d - 1) * n:]), "quotient has more than d - 1 pieces: the circuit degree is too small"
    h_coeff = h_coeff[:(d - 1) * n] + [0] * max(0, (d - 1) * n - len(h_coeff))
    pieces = [h_coeff[i * n:(i + 1) * n] for i in range(d - 1)]
    for p_ in pieces:
        tr.write_point(srs.commit(p_))
    x = tr.squeeze()

    # ---- evaluations, in proof order; `queries` collects the multi-open's (polynomial, point, eval)
    point = lambda rot: x * pow(omega, rot % n, R) % R

    def ev(cf, rot, write=True):
        e = b.eval_polynomial(cf, point(rot))
        if write:
            tr.write_scalar(e)
        return e
    adv_evals = [ev(coeff[(ADVICE, i)], rot) for i, rot in circ.advice_queries]
    fix_evals = [ev(coeff[(FIXED, i)], rot) for i, rot in circ.fixed_queries]
    random_eval = ev(random_coeff, 0)
    sigma_evals = [ev(sig_coeff[j], 0) for j in range(Pn)]
    z_evals = []
    for c in range(C):
        e0, e1 = ev(pz_coeff[c], 0), ev(pz_coeff[c], 1)
        z_evals.append((e0, e1, ev(pz_coeff[c], rot_last) if c + 1 < C else None))
    lk_evals = [(ev(phi_coeff[l], 0), ev(phi_coeff[l], 1), ev(m_coeff[l], 0)) for l in range(L)]
    xn = pow(x, n, R)
    hcomb = [0] * n
    for p_ in reversed(pieces):
        hcomb = [(a * xn + c_) % R for a, c_ in zip(hcomb, p_)]
    h_eval = ev(hcomb, 0, write=False)          # the verifier derives this evaluation itself

    # halo2's order: advice, permutation products (x, wx per set, then w^last x in reverse set order),
    # lookups, fixed, permutation sigma, h, random polynomial
    queries = [(coeff[(ADVICE, i)], point(rot), e) for (i, rot), e in zip(circ.advice_queries, adv_evals)]
    for c in range(C):
        queries += [(pz_coeff[c], point(0), z_evals[c][0]), (pz_coeff[c], point(1), z_evals[c][1])]
    for c in reversed(range(C - 1)):
        queries.append((pz_coeff[c], point(rot_last), z_evals[c][2]))
    for l in range(L):
        queries += [(phi_coeff[l], point(0), lk_evals[l][0]), (phi_coeff[l], point(1), lk_evals[l][1]), (m_coeff[l], point(0), lk_evals[l][2])]
    queries += [(coeff[(FIXED, i)], point(rot), e) for (i, rot), e in zip(circ.fixed_queries, fix_evals)]
    queries += [(sig_coeff[j], point(0), sigma_evals[j]) for j in range(Pn)]
    queries += [(hcomb, point(0), h_eval), (random_coeff, point(0), random_eval)]

    def lincomb(polys, ch):               # sum_j ch^j * polys[j]
        acc_, pw = [0] * n, 1
        for p_ in polys:
            acc_ = [(a + pw * c_) % R for a, c_ in zip(acc_, p_)]
            pw = pw * ch % R
        return acc_

    if multiopen == "gwc":
        # poly::kzg::multiopen::gwc::ProverGWC::create_proof
        v = tr.squeeze()
        groups = []
        for cf, pt, _ in queries:
            for g_ in groups:
                if g_[0] == pt:
                    g_[1].append(cf)
                    break
            else:
                groups.append([pt, [cf]])
        for pt, polys in groups:
            tr.write_point(srs.commit(b.kate_division(lincomb(polys, v), pt)))
        return bytes(tr.proof)

    # ---- SHPLONK (BDFG21): poly::kzg::multiopen::shplonk::ProverSHPLONK::create_proof
    yy = tr.squeeze()
    sets, super_points, eval_of = rotation_sets(queries)
    v = tr.squeeze()
    low = []          # per set, per polynomial: r_ij(X), the interpolation of its evaluations over the set's points
    quot = []
    for points, members in sets:
        rs = [newton_interpolate(points, [eval_of(cf, p_) for p_ in points]) for cf in members]
        numer = lincomb([[(c_ - (r_[t] if t < len(r_) else 0)) % R for t, c_ in enumerate(cf)] for cf, r_ in zip(members, rs)], yy)
        for z in points:
            numer = b.kate_division(numer, z)
        quot.append(numer + [0] * (n - len(numer)))
        low.append(rs)
    h_x = lincomb(quot, v)
    tr.write_point(srs.commit(h_x))
    uu = tr.squeeze()
    l_x, z_diffs, vpow = [0] * n, [], 1
    for (points, members), rs in zip(sets, low):
        z_i = product_of_differences(uu, [p_ for p_ in super_points if p_ not in points])
        z_diffs.append(z_i)
        inner = lincomb([[(c_ - (b.eval_polynomial(r_, uu) if t == 0 else 0)) % R for t, c_ in enumerate(cf)] for cf, r_ in zip(members, rs)], yy)
        l_x = [(a + vpow * z_i % R * c_) % R for a, c_ in zip(l_x, inner)]
        vpow = vpow * v % R
    zt_eval = product_of_differences(uu, super_points)
    l_x = [(a - zt_eval * c_) % R for a, c_ in zip(l_x, h_x)]
    assert b.eval_polynomial(l_x, uu) == 0
    z0_inv = b.fr_inv(z_diffs[0])
    tr.write_point(srs.commit([c_ * z0_inv % R for c_ in b.kate_division(l_x, uu)]))
    return bytes(tr.proof)
