"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- verifier for proofs produced by csrc/prover.hip.

Restates the verifier half of halo2_proofs (plonk::verify_proof with the GWC multi-open and a
Blake2b transcript) for the protocol the HIP prover implements, with
a real pairing check (oracle/pairing.py).  The reference keeps this half on the CPU and uses it as
its only acceptance criterion for the hot path [REF circuit-benchmarks/src/super_circuit.rs:141-154];
the Rust verifier itself cannot be built here, so acceptance by THIS verifier is tier
T0 of SURVEY 8c, not T1.

Also provides `check_witness`, a MockProver-style row-by-row constraint check used to make sure
the test circuits are satisfied before they are proved.
"""
from __future__ import annotations

import hashlib
from typing import Dict, List, Optional, Sequence, Tuple

from . import bn254 as b
from . import pairing as pr

R, P = b.R_MOD, b.P_MOD
FIXED, ADVICE, INSTANCE = 0, 1, 2
Q_PUSH_COL, Q_PUSH_CONST, Q_ADD, Q_SUB, Q_MUL, Q_NEG = 1, 2, 3, 4, 5, 6


# ------------------------------------------------------------------------------------ transcript
class Blake2bRead:
    """halo2_proofs::transcript::Blake2bRead + Challenge255."""

    def __init__(self, proof: bytes):
        self.h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.proof, self.pos = proof, 0

    def common_point(self, pt):
        if pt is None:
            self.h.update(b"\x01" + bytes(64))
        else:
            self.h.update(b"\x01" + pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little"))

    def common_scalar(self, s: int):
        self.h.update(b"\x02" + (s % R).to_bytes(32, "little"))

    def read_point(self):
        raw = self.proof[self.pos:self.pos + 32]
        assert len(raw) == 32, "proof truncated"
        self.pos += 32
        pt = decompress_g1(raw)
        self.common_point(pt)
        return pt

    def read_scalar(self) -> int:
        raw = self.proof[self.pos:self.pos + 32]
        assert len(raw) == 32, "proof truncated"
        self.pos += 32
        s = int.from_bytes(raw, "little")
        assert s < R, "non-canonical scalar"
        self.common_scalar(s)
        return s

    def squeeze(self) -> int:
        self.h.update(b"\x00")
        return b.fr_from_uniform_bytes(self.h.copy().digest())


def decompress_g1(raw: bytes):
    if raw == bytes(32):
        return None
    v = int.from_bytes(raw, "little")
    sign, x = v >> 255, v & ((1 << 255) - 1)
    assert x < P
    y2 = (x * x * x + 3) % P
    y = pow(y2, (P + 1) // 4, P)
    assert y * y % P == y2, "point not on curve"
    if (y & 1) != sign:
        y = P - y
    return (x, y)


# ------------------------------------------------------------------------------------ expressions
C_CHAL0 = 0xFFFD0000


class Consts:
    """user constants + (once known) the user challenges behind the abstract references"""

    def __init__(self, consts, challenges=None):
        self.consts, self.challenges = consts, challenges

    def __getitem__(self, a):
        if a >= C_CHAL0:
            assert self.challenges is not None, "challenge used where none is available"
            return self.challenges[a - C_CHAL0]
        return self.consts[a]


def eval_program(prog, lookup_col, consts) -> int:
    st: List[int] = []
    for op, a, bb in prog:
        if op == Q_PUSH_COL:
            rot = bb if bb < (1 << 31) else bb - (1 << 32)
            st.append(lookup_col(a >> 24, a & 0xFFFFFF, rot))
        elif op == Q_PUSH_CONST:
            st.append(consts[a])
        elif op == Q_NEG:
            st[-1] = (-st[-1]) % R
        else:
            y = st.pop()
            x = st[-1]
            st[-1] = (x + y) % R if op == Q_ADD else ((x - y) % R if op == Q_SUB else x * y % R)
    assert len(st) == 1
    return st[0]


def compress(vals: Sequence[int], theta: int) -> int:
    acc = 0
    for v in vals:
        acc = (acc * theta + v) % R
    return acc


def check_witness(circ, advice: Sequence[Sequence[int]], instance: Sequence[Sequence[int]], challenges=None) -> Optional[str]:
    """MockProver-style check over the usable rows; returns None or a description of the failure."""
    n, u = circ.n, circ.u
    consts = Consts(circ.consts, challenges)
    cols = {FIXED: circ.fixed, ADVICE: advice, INSTANCE: [list(c) + [0] * (circ.n - len(c)) for c in instance]}
    gates = [circ.compile(g) for g in circ.gates]
    lookups = [([circ.compile(e) for e in i], [circ.compile(e) for e in t]) for i, t in circ.lookups]
    for row in range(u):
        look = lambda t, i, rot: cols[t][i][(row + rot) % n]
        for gi, g in enumerate(gates):
            if eval_program(g, look, consts) != 0:
                return f"gate {gi} not satisfied at row {row}"
    for a, c in circ.copies:
        if cols[a[0]][a[1]][a[2]] != cols[c[0]][c[1]][c[2]]:
            return f"copy constraint {a} == {c} violated"
    for li, (ins, tabs) in enumerate(lookups):
        table = set()
        for row in range(u):
            look = lambda t, i, rot: cols[t][i][(row + rot) % n]
            table.add(tuple(eval_program(p, look, consts) for p in tabs))
        for row in range(u):
            look = lambda t, i, rot: cols[t][i][(row + rot) % n]
            if tuple(eval_program(p, look, consts) for p in ins) not in table:
                return f"lookup {li}: input at row {row} not in table"
    return None


# ------------------------------------------------------------------------------------ verifier
def _queries(circ):
    adv, fix = [], []

    def scan(prog):
        for op, a, bb in prog:
            if op != Q_PUSH_COL:
                continue
            t, i = a >> 24, a & 0xFFFFFF
            rot = bb if bb < (1 << 31) else bb - (1 << 32)
            dst = adv if t == ADVICE else (fix if t == FIXED else None)
            if dst is not None and (i, rot) not in dst:
                dst.append((i, rot))
    for g in circ.gates:
        scan(circ.compile(g))
    for ins, tabs in circ.lookups:
        for e in ins:
            scan(circ.compile(e))
        for e in tabs:
            scan(circ.compile(e))
    for t, i in circ.perm_cols:
        scan([(Q_PUSH_COL, (t << 24) | i, 0)])
    return adv, fix


def _interpolate(xs, ys):
    """coefficients (low first) of the polynomial through (xs[i], ys[i])"""
    m = len(xs)
    out = [0] * m
    for a in range(m):
        num, den = [1], 1
        for c in range(m):
            if c == a:
                continue
            nx = [0] * (len(num) + 1)
            for t, v in enumerate(num):
                nx[t + 1] = (nx[t + 1] + v) % R
                nx[t] = (nx[t] - v * xs[c]) % R
            num = nx
            den = den * (xs[a] - xs[c]) % R
        sc = ys[a] * b.fr_inv(den) % R
        for t, v in enumerate(num):
            out[t] = (out[t] + v * sc) % R
    return out


def _verify_shplonk(tr, proof, opens, rots, point, s_g2) -> bool:
    """SHPLONK / BDFG21 verifier for the prover's variant: sets in order of first
    appearance, Horner powers of y inside a set and of v across sets."""
    y, v = tr.squeeze(), tr.squeeze()
    polys = []          # (commitment, [rots], [evals])
    for com, rot, e in opens:
        for p in polys:
            if p[0] is com:
                p[1].append(rot); p[2].append(e)
                break
        else:
            polys.append((com, [rot], [e]))
    sets = []           # (sorted rots, [poly indices])
    for pi, p in enumerate(polys):
        key = sorted(p[1])
        for s_ in sets:
            if s_[0] == key:
                s_[1].append(pi)
                break
        else:
            sets.append((key, [pi]))
    h_com = tr.read_point()
    u = tr.squeeze()
    pi_com = tr.read_point()
    if tr.pos != len(proof):
        return False
    zT = 1
    for r_ in rots:
        zT = zT * (u - point(r_)) % R
    L, cpow = None, 1
    for key, members in reversed(sets):
        zs = [point(r_) for r_ in key]
        qcom, Ru = None, 0
        for pi in members:
            com, prots, pevals = polys[pi]
            ys = [pevals[prots.index(r_)] for r_ in key]
            rj = _interpolate(zs, ys)
            rju = 0
            for c in reversed(rj):
                rju = (rju * u + c) % R
            qcom = b.g1_add(b.g1_mul(qcom, y) if qcom is not None else None, com)
            Ru = (Ru * y + rju) % R
        zt = 1
        for r_ in rots:
            if r_ not in key:
                zt = zt * (u - point(r_)) % R
        coef = cpow * zt % R
        term = b.g1_add(qcom, b.g1_neg(b.g1_mul(b.G1_GEN, Ru)))
        L = b.g1_add(L, b.g1_mul(term, coef))
        cpow = cpow * v % R
    L = b.g1_add(L, b.g1_neg(b.g1_mul(h_com, zT)))
    # L(X) = (X - u) pi(X):  e(L + u pi, G2) == e(pi, [s]G2)
    lhs = b.g1_add(L, b.g1_mul(pi_com, u))
    return pr.pairing_check([(lhs, pr.ec_neg(pr.G2_GEN)), (pi_com, s_g2)])


def verify(circ, vk_commitments: Sequence, vk_repr: int, instance: Sequence[Sequence[int]], proof: bytes, s_g2, multiopen: str = "gwc") -> bool:
    """vk_commitments: F fixed then P sigma affine points (int tuples); s_g2: [s]G2 of the SRS."""
    n, k, u, bf, d = circ.n, circ.k, circ.u, circ.bf, circ.degree()
    F, A, I, Pn, L = circ.F, circ.A, circ.I, len(circ.perm_cols), len(circ.lookups)
    chunk = d - 2
    C = (Pn + chunk - 1) // chunk if Pn else 0
    omega = b.omega_for_k(k)
    gates = [circ.compile(g) for g in circ.gates]
    lookups = [([circ.compile(e) for e in i], [circ.compile(e) for e in t]) for i, t in circ.lookups]
    adv_q, fix_q = _queries(circ)
    fixed_com, sigma_com = list(vk_commitments[:F]), list(vk_commitments[F:F + Pn])

    tr = Blake2bRead(proof)
    tr.common_scalar(vk_repr)
    for col in instance:          # exactly the values given, as halo2's verify_proof does (an n-row image stands for all usable rows)
        if u < len(col) < n:
            return False
        for row in range(min(len(col), u)):
            tr.common_scalar(col[row])
    # advice commitments phase by phase, each followed by that phase's challenges
    adv_com = [None] * A
    adv_phase = getattr(circ, "advice_phase", [0] * A)
    chal_phase = getattr(circ, "challenge_phase", [])
    challenges = [0] * len(chal_phase)
    for ph in range(max([0] + list(adv_phase) + list(chal_phase)) + 1):
        for i in range(A):
            if adv_phase[i] == ph:
                adv_com[i] = tr.read_point()
        for ci, cp in enumerate(chal_phase):
            if cp == ph:
                challenges[ci] = tr.squeeze()
    consts = Consts(circ.consts, challenges)
    theta = tr.squeeze()
    m_com = [tr.read_point() for _ in range(L)]
    beta, gamma = tr.squeeze(), tr.squeeze()
    z_com = [tr.read_point() for _ in range(C)]
    phi_com = [tr.read_point() for _ in range(L)]
    random_com = tr.read_point()
    y = tr.squeeze()
    h_com = [tr.read_point() for _ in range(d - 1)]
    x = tr.squeeze()

    rot_last = -(bf + 1)
    point = lambda rot: x * pow(omega, rot % n, R) % R
    opens: List[Tuple[object, int, int]] = []   # (commitment, rot, eval)
    ev: Dict[Tuple[int, int, int], int] = {}
    for i, rot in adv_q:
        e = tr.read_scalar(); ev[(ADVICE, i, rot)] = e; opens.append((adv_com[i], rot, e))
    for i, rot in fix_q:
        e = tr.read_scalar(); ev[(FIXED, i, rot)] = e; opens.append((fixed_com[i], rot, e))
    random_eval = tr.read_scalar(); opens.append((random_com, 0, random_eval))
    sigma_eval = []
    for j in range(Pn):
        e = tr.read_scalar(); sigma_eval.append(e); opens.append((sigma_com[j], 0, e))
    z_eval = []
    for c in range(C):
        e0 = tr.read_scalar(); opens.append((z_com[c], 0, e0))
        e1 = tr.read_scalar(); opens.append((z_com[c], 1, e1))
        el = None
        if c + 1 < C:
            el = tr.read_scalar(); opens.append((z_com[c], rot_last, el))
        z_eval.append((e0, e1, el))
    lk_eval = []
    for l in range(L):
        p0 = tr.read_scalar(); opens.append((phi_com[l], 0, p0))
        p1 = tr.read_scalar(); opens.append((phi_com[l], 1, p1))
        me = tr.read_scalar(); opens.append((m_com[l], 0, me))
        lk_eval.append((p0, p1, me))

    # ---- instance evaluations are computed by the verifier (KZG: QUERY_INSTANCE = false)
    xn = pow(x, n, R)
    def lagrange_at(i: int, pt: int) -> int:    # L_i(pt)
        wi = pow(omega, i, R)
        return wi * (pow(pt, n, R) - 1) % R * b.fr_inv(n * (pt - wi) % R) % R
    def col_eval(t, i, rot):
        if t == INSTANCE:
            pt = point(rot)
            return sum(instance[i][row] * lagrange_at(row, pt) for row in range(min(len(instance[i]), u)) if instance[i][row]) % R
        return ev[(t, i, rot)]

    l0 = lagrange_at(0, x)
    l_last = lagrange_at(u, x)
    l_blind = sum(lagrange_at(i, x) for i in range(u + 1, n)) % R
    l_active = (1 - l_last - l_blind) % R

    # ---- expected numerator: same constraint order as the prover's quotient program
    acc = 0
    def fold(term):
        nonlocal acc
        acc = (acc * y + term) % R
    for g in gates:
        fold(eval_program(g, col_eval, consts))
    if C:
        fold(l0 * (1 - z_eval[0][0]) % R)
        zl = z_eval[C - 1][0]
        fold(l_last * (zl * zl - zl) % R)
        for c in range(1, C):
            fold(l0 * (z_eval[c][0] - z_eval[c - 1][2]) % R)
        for c in range(C):
            left, right = z_eval[c][1], z_eval[c][0]
            for j in range(c * chunk, min(Pn, (c + 1) * chunk)):
                t, i = circ.perm_cols[j]
                v = col_eval(t, i, 0)
                left = left * ((v + beta * sigma_eval[j] + gamma) % R) % R
                right = right * ((v + beta * pow(b.FR_DELTA, j, R) % R * x + gamma) % R) % R
            fold(l_active * (left - right) % R)
    for l, (ins, tabs) in enumerate(lookups):
        p0, p1, me = lk_eval[l]
        f = compress([eval_program(p, col_eval, consts) for p in ins], theta)
        t = compress([eval_program(p, col_eval, consts) for p in tabs], theta)
        fold(l0 * p0 % R)
        fold(l_last * p0 % R)
        fold(l_active * (((p1 - p0) * (f + beta) % R * (t + beta) - ((t + beta) - me * (f + beta))) % R) % R)
    h_eval = acc * b.fr_inv((xn - 1) % R) % R
    # commitment to h(X) = sum_i x^(n i) h_i(X)
    hc = None
    for com in reversed(h_com):
        hc = b.g1_add(b.g1_mul(hc, xn) if hc is not None else None, com)
    opens.append((hc, 0, h_eval))

    rots = []
    for _, rot, _ in opens:
        if rot not in rots:
            rots.append(rot)
    if multiopen == "shplonk":
        return _verify_shplonk(tr, proof, opens, rots, point, s_g2)
    # ---- GWC: one witness per distinct point, in order of first appearance
    v = tr.squeeze()
    witnesses = [tr.read_point() for _ in rots]
    if tr.pos != len(proof):
        return False
    uch = tr.squeeze()
    lhs, rhs, upow = None, None, 1
    for rot, W in zip(rots, witnesses):
        cb, eb = None, 0
        for com, r_, e in opens:
            if r_ != rot:
                continue
            cb = b.g1_add(b.g1_mul(cb, v) if cb is not None else None, com)
            eb = (eb * v + e) % R
        z = point(rot)
        term = b.g1_add(b.g1_add(cb, b.g1_neg(b.g1_mul(b.G1_GEN, eb))), b.g1_mul(W, z))
        lhs = b.g1_add(lhs, b.g1_mul(term, upow))
        rhs = b.g1_add(rhs, b.g1_mul(W, upow))
        upow = upow * uch % R
    # e(lhs, G2) == e(rhs, [s]G2)
    return pr.pairing_check([(lhs, pr.ec_neg(pr.G2_GEN)), (rhs, s_g2)])
