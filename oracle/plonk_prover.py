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
from .plonk_verifier import (ADVICE, FIXED, INSTANCE, Consts, _interpolate, _queries, compress, eval_program)

R = b.R_MOD
Q_PUSH_COL = 1


# ------------------------------------------------------------------------------------ transcript / RNG
class Blake2bWrite:
    """halo2_proofs::transcript::Blake2bWrite + Challenge255."""

    def __init__(self):
        self.h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.proof = bytearray()

    def common_point(self, pt):
        self.h.update(b"\x01" + (bytes(64) if pt is None else pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little")))

    def common_scalar(self, s: int):
        self.h.update(b"\x02" + (s % R).to_bytes(32, "little"))

    def write_point(self, pt):
        self.common_point(pt)
        self.proof += b.g1_compress(pt)

    def write_scalar(self, s: int):
        self.common_scalar(s)
        self.proof += (s % R).to_bytes(32, "little")

    def squeeze(self) -> int:
        self.h.update(b"\x00")
        return b.fr_from_uniform_bytes(self.h.copy().digest())


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
                 seed16: bytes = bytes(16), multiopen: str = "gwc") -> bytes:
    n, k, u, bf, d = circ.n, circ.k, circ.u, circ.bf, circ.degree()
    A, Pn, L = circ.A, len(circ.perm_cols), len(circ.lookups)
    chunk = d - 2
    C = (Pn + chunk - 1) // chunk if Pn else 0
    dom = b.EvaluationDomain(d, k)
    ext_k, ne = dom.extended_k, 1 << dom.extended_k
    step = ne // n
    omega = dom.omega
    gates = [circ.compile(g) for g in circ.gates]
    lookups = [([circ.compile(e) for e in i], [circ.compile(e) for e in t]) for i, t in circ.lookups]
    adv_q, fix_q = _queries(circ)
    sigma = circ.sigma_columns()
    rng, tr = XorShiftRng(seed16), Blake2bWrite()

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
    # ---- lookups, round 1: multiplicities
    lk_f, lk_t, lk_m = [], [], []
    for ins, tabs in lookups:
        f = [compress([eval_program(p, row_lookup(r), consts) for p in ins], theta) for r in range(n)]
        t = [compress([eval_program(p, row_lookup(r), consts) for p in tabs], theta) for r in range(n)]
        first = {}
        for r in range(u):
            first.setdefault(t[r], r)
        m = [0] * n
        for r in range(u):
            assert f[r] in first, f"lookup input at row {r} is not in the table"
            m[first[f[r]]] += 1
        for r in range(u + 1, n):
            m[r] = rng.next_fr()
        lk_f.append(f); lk_t.append(t); lk_m.append(m)
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
    # ---- lookups, round 2: grand sums
    lk_phi = []
    for f, t, m in zip(lk_f, lk_t, lk_m):
        phi = [0] * n
        for r in range(n - 1):
            g_ = (b.fr_inv((f[r] + beta) % R) - m[r] * b.fr_inv((t[r] + beta) % R)) % R
            phi[r + 1] = (phi[r] + g_) % R
        assert phi[u] == 0, "lookup grand sum does not close"
        for r in range(n - bf, n):
            phi[r] = rng.next_fr()
        lk_phi.append(phi)
    for phi in lk_phi:
        tr.write_point(srs.commit_lagrange(phi))
    # ---- vanishing argument: blinding polynomial from ChaCha20 in counter mode
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
        for l, (ins, tabs) in enumerate(lookups):
            p0, p1, me = at(phi_ext[l]), at(phi_ext[l], 1), at(m_ext[l])
            f = compress([eval_program(p, col_at, consts) for p in ins], theta)
            t = compress([eval_program(p, col_at, consts) for p in tabs], theta)
            acc = (acc * y + l0_e[j] * p0) % R
            acc = (acc * y + ll_e[j] * p0) % R
            acc = (acc * y + la_e[j] * ((p1 - p0) * (f + beta) % R * (t + beta) - ((t + beta) - me * (f + beta)))) % R
        h_ext[j] = acc * dom.t_evaluations[j % len(dom.t_evaluations)] % R
    h_coeff = dom.extended_to_coeff(h_ext)
    h_coeff += [0] * ((d - 1) * n - len(h_coeff))
    pieces = [h_coeff[i * n:(i + 1) * n] for i in range(d - 1)]
    for p_ in pieces:
        tr.write_point(srs.commit(p_))
    x = tr.squeeze()

    # ---- evaluations, in proof order
    point = lambda rot: x * pow(omega, rot % n, R) % R
    opens = []          # (coefficients, rot, eval)

    def open_(cf, rot, write=True):
        e = b.eval_polynomial(cf, point(rot))
        opens.append((cf, rot, e))
        if write:
            tr.write_scalar(e)
    for i, rot in adv_q:
        open_(coeff[(ADVICE, i)], rot)
    for i, rot in fix_q:
        open_(coeff[(FIXED, i)], rot)
    open_(random_coeff, 0)
    for j in range(Pn):
        open_(sig_coeff[j], 0)
    for c in range(C):
        open_(pz_coeff[c], 0)
        open_(pz_coeff[c], 1)
        if c + 1 < C:
            open_(pz_coeff[c], rot_last)
    for l in range(L):
        open_(phi_coeff[l], 0)
        open_(phi_coeff[l], 1)
        open_(m_coeff[l], 0)
    xn = pow(x, n, R)
    hcomb = [0] * n
    for p_ in reversed(pieces):
        hcomb = [(a * xn + c_) % R for a, c_ in zip(hcomb, p_)]
    open_(hcomb, 0, write=False)          # the verifier derives this evaluation itself
    rots: List[int] = []
    for _, rot, _ in opens:
        if rot not in rots:
            rots.append(rot)

    def lincomb(polys, ch):               # Horner: ((p0 * ch + p1) * ch + p2) ...
        acc_ = [0] * n
        for p_ in polys:
            acc_ = [(a * ch + c_) % R for a, c_ in zip(acc_, p_)]
        return acc_

    if multiopen == "gwc":
        v = tr.squeeze()
        for rot in rots:
            batch = lincomb([cf for cf, r_, _ in opens if r_ == rot], v)
            tr.write_point(srs.commit(b.kate_division(batch, point(rot))))
        return bytes(tr.proof)

    # ---- SHPLONK (BDFG21), the prover side of plonk_verifier._verify_shplonk
    yy, v = tr.squeeze(), tr.squeeze()
    polys = []          # [coefficients, [rots], [evals]] by identity of the coefficient list
    for cf, rot, e in opens:
        for p_ in polys:
            if p_[0] is cf:
                p_[1].append(rot); p_[2].append(e)
                break
        else:
            polys.append([cf, [rot], [e]])
    sets = []
    for pi, p_ in enumerate(polys):
        key_ = sorted(p_[1])
        for s_ in sets:
            if s_[0] == key_:
                s_[1].append(pi)
                break
        else:
            sets.append((key_, [pi]))
    qfull, hset, Rset = [], [], []
    for key_, members in sets:
        zs = [point(r_) for r_ in key_]
        Rk = [0] * len(key_)
        for pi in members:
            cf, prots, pevals = polys[pi]
            rj = _interpolate(zs, [pevals[prots.index(r_)] for r_ in key_])
            Rk = [(a * yy + c_) % R for a, c_ in zip(Rk, rj)]
        qf = lincomb([polys[pi][0] for pi in members], yy)
        hs = list(qf)
        for t_ in range(len(Rk)):
            hs[t_] = (hs[t_] - Rk[t_]) % R
        for z in zs:
            hs = b.kate_division(hs, z)
        hs += [0] * (n - len(hs))
        qfull.append(qf); hset.append(hs); Rset.append(Rk)
    hpoly = lincomb(hset, v)
    tr.write_point(srs.commit(hpoly))
    uu = tr.squeeze()
    zT = 1
    for r_ in rots:
        zT = zT * (uu - point(r_)) % R
    Lx = [0] * n
    constant, cpow = 0, 1
    for si in reversed(range(len(sets))):
        zt = 1
        for r_ in rots:
            if r_ not in sets[si][0]:
                zt = zt * (uu - point(r_)) % R
        coef = cpow * zt % R
        Lx = [(a + coef * c_) % R for a, c_ in zip(Lx, qfull[si])]
        constant = (constant + coef * b.eval_polynomial(Rset[si], uu)) % R
        cpow = cpow * v % R
    Lx = [(a - zT * c_) % R for a, c_ in zip(Lx, hpoly)]
    Lx[0] = (Lx[0] - constant) % R
    tr.write_point(srs.commit(b.kate_division(Lx, uu)))
    return bytes(tr.proof)
