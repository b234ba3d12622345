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
from .transcripts import Blake2b as Blake2bRead, decompress_g1, make as make_transcript  # noqa: E402,F401


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
    lookups = [([circ.compile(e) for e in lk.table], [[circ.compile(e) for e in i] for i in lk.inputs]) for lk in circ.lookups]
    for row in range(u):
        look = lambda t, i, rot: cols[t][i][(row + rot) % n]
        for gi, g in enumerate(gates):
            if eval_program(g, look, consts) != 0:
                return f"gate {gi} not satisfied at row {row}"
    for a, c in circ.copies:
        if cols[a[0]][a[1]][a[2]] != cols[c[0]][c[1]][c[2]]:
            return f"copy constraint {a} == {c} violated"
    for li, (tabs, inputs) in enumerate(lookups):
        table = set()
        for row in range(u):
            look = lambda t, i, rot: cols[t][i][(row + rot) % n]
            table.add(tuple(eval_program(p, look, consts) for p in tabs))
        for ii, ins in enumerate(inputs):
            for row in range(u):
                look = lambda t, i, rot: cols[t][i][(row + rot) % n]
                if tuple(eval_program(p, look, consts) for p in ins) not in table:
                    return f"lookup {li}, input set {ii}: input at row {row} not in table"
    return None


MOCK_GATE, MOCK_LOOKUP, MOCK_PERMUTATION = 1, 2, 3


def mock_failures(circ, advice: Sequence[Sequence[int]], instance: Sequence[Sequence[int]], challenges=None, gate_rows=None, lookup_rows=None):
    """halo2 ``dev::MockProver::verify_at_rows_par`` restated (``dev.rs``: gates on `gate_row_ids`, lookups on
    `lookup_input_row_ids`, the permutation over the whole mapping), as the sorted list of
    (kind, index, sub, row) that include/zkmi355.h:zk_mock_verify reports:
      (1, gate polynomial, 0, row)              VerifyFailure::ConstraintNotSatisfied
      (2, lookup argument, input tuple, row)    VerifyFailure::Lookup (tuples compared exactly, table = its usable rows)
      (3, permutation column, 0, row)           VerifyFailure::Permutation (cell differs from the cell the mapping names)
    Row ids default to every usable row (``verify_par``)."""
    n, u = circ.n, circ.u
    consts = Consts(circ.consts, challenges)
    cols = {FIXED: circ.fixed, ADVICE: advice, INSTANCE: [list(c) + [0] * (circ.n - len(c)) for c in instance]}
    gates = [circ.compile(g) for g in circ.gates]
    lookups = [([circ.compile(e) for e in lk.table], [[circ.compile(e) for e in i] for i in lk.inputs]) for lk in circ.lookups]
    gate_rows = range(u) if gate_rows is None else gate_rows
    lookup_rows = range(u) if lookup_rows is None else lookup_rows
    assert all(0 <= r < u for r in gate_rows) and all(0 <= r < u for r in lookup_rows), "row ids must be usable rows"
    out = []

    def looker(row):
        return lambda t, i, rot: cols[t][i][(row + rot) % n]

    for row in gate_rows:
        look = looker(row)
        for gi, g in enumerate(gates):
            if eval_program(g, look, consts) != 0:
                out.append((MOCK_GATE, gi, 0, row))
    for li, (tabs, inputs) in enumerate(lookups):
        table = set()
        for row in range(u):
            look = looker(row)
            table.add(tuple(eval_program(p, look, consts) for p in tabs))
        for ii, ins in enumerate(inputs):
            for row in lookup_rows:
                look = looker(row)
                if tuple(eval_program(p, look, consts) for p in ins) not in table:
                    out.append((MOCK_LOOKUP, li, ii, row))
    mapping = circ.permutation_mapping()
    for j, (t, c) in enumerate(circ.perm_cols):
        for i in range(n):
            j2, i2 = mapping[j][i]
            t2, c2 = circ.perm_cols[j2]
            if cols[t][c][i] != cols[t2][c2][i2]:
                out.append((MOCK_PERMUTATION, j, 0, i))
    return sorted(out)


# ------------------------------------------------------------------------------------ verifier
def _interpolate(xs, ys):
    """halo2 arithmetic::lagrange_interpolate: coefficients (low first) of the polynomial through (xs[i], ys[i])"""
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


class Com:
    """one committed polynomial as the multi-open sees it (halo2 `CommitmentReference`: queries on the
    same polynomial are recognised by identity, not by value)"""

    def __init__(self, point):
        self.point = point


def shplonk_sets(queries):
    """halo2 poly::kzg::multiopen::shplonk::construct_intermediate_sets: polynomials in order of first
    appearance, each with the set of points it is opened at; rotation sets in order of first
    appearance, each listing its polynomials.  queries: (object, point, eval)."""
    polys, point_sets = [], []
    for obj, pt, _ in queries:
        for i, o in enumerate(polys):
            if o is obj:
                if pt not in point_sets[i]:
                    point_sets[i].append(pt)
                break
        else:
            polys.append(obj)
            point_sets.append([pt])
    sets = []           # [sorted points, [polynomial objects]]
    for obj, pts in zip(polys, point_sets):
        key = sorted(pts)                  # BTreeSet<F>: ordered by canonical value
        for s_ in sets:
            if s_[0] == key:
                s_[1].append(obj)
                break
        else:
            sets.append([key, [obj]])
    super_points = sorted({pt for _, pt, _ in queries})

    def eval_of(obj, pt):
        return next(e for o, p_, e in queries if o is obj and p_ == pt)
    return sets, super_points, eval_of


def _vanishing_at(points, u):
    acc = 1
    for z in points:
        acc = acc * (u - z) % R
    return acc


def _verify_shplonk(tr, queries, s_g2) -> bool:
    """halo2 poly::kzg::multiopen::shplonk::VerifierSHPLONK::verify_proof (BDFG21): ascending powers of
    y inside a rotation set and of v across the sets, everything normalised by the first set's Z_{T \\ S_0}(u)."""
    sets, super_points, eval_of = shplonk_sets(queries)
    y, v = tr.squeeze(), tr.squeeze()
    h1 = tr.read_point()
    u = tr.squeeze()
    h2 = tr.read_point()
    if not tr.exhausted():
        return False
    outer, r_outer, z_0, z_0_diff_inv, vpow = None, 0, 0, 0, 1
    for i, (points, members) in enumerate(sets):
        z_diff = _vanishing_at([p_ for p_ in super_points if p_ not in points], u)
        if i == 0:
            z_0 = _vanishing_at(points, u)
            z_0_diff_inv = b.fr_inv(z_diff)
            z_diff = 1
        else:
            z_diff = z_diff * z_0_diff_inv % R
        inner, r_inner, ypow = None, 0, 1
        for obj in members:
            r_x = _interpolate(points, [eval_of(obj, p_) for p_ in points])
            r_inner = (r_inner + ypow * b.eval_polynomial(r_x, u)) % R
            inner = b.g1_add(inner, b.g1_mul(obj.point, ypow))
            ypow = ypow * y % R
        outer = b.g1_add(outer, b.g1_mul(inner, vpow * z_diff % R))
        r_outer = (r_outer + vpow * r_inner % R * z_diff) % R
        vpow = vpow * v % R
    outer = b.g1_add(outer, b.g1_neg(b.g1_mul(b.G1_GEN, r_outer)))
    outer = b.g1_add(outer, b.g1_neg(b.g1_mul(h1, z_0)))
    outer = b.g1_add(outer, b.g1_mul(h2, u))
    # e(h2, [s]) == e(outer, [1])
    return pr.pairing_check([(outer, pr.ec_neg(pr.G2_GEN)), (h2, s_g2)])


def _verify_gwc(tr, queries, s_g2) -> bool:
    """halo2 poly::kzg::multiopen::gwc::VerifierGWC::verify_proof: queries grouped by point in order of
    first appearance, ascending powers of v inside a group and of u across the groups."""
    v = tr.squeeze()
    groups = []          # [point, [(object, eval)]]
    for obj, pt, e in queries:
        for g_ in groups:
            if g_[0] == pt:
                g_[1].append((obj, e))
                break
        else:
            groups.append([pt, [(obj, e)]])
    witnesses = [tr.read_point() for _ in groups]
    if not tr.exhausted():
        return False
    u = tr.squeeze()
    left, right, upow = None, None, 1
    for (z, members), w in zip(groups, witnesses):
        cb, eb, vpow = None, 0, 1
        for obj, e in members:
            cb = b.g1_add(cb, b.g1_mul(obj.point, vpow))
            eb = (eb + vpow * e) % R
            vpow = vpow * v % R
        term = b.g1_add(b.g1_add(cb, b.g1_neg(b.g1_mul(b.G1_GEN, eb))), b.g1_mul(w, z))
        right = b.g1_add(right, b.g1_mul(term, upow))
        left = b.g1_add(left, b.g1_mul(w, upow))
        upow = upow * u % R
    # e(left, [s]) == e(right, [1])
    return pr.pairing_check([(right, pr.ec_neg(pr.G2_GEN)), (left, s_g2)])


def default_vk_repr(circ, vk_commitments) -> int:
    """zk_pk_create's stand-in for halo2's `vk.transcript_repr` when the caller supplies none:
    Blake2b-512 (personal "Halo2-Verify-Key") over the constraint-system part of the key blob and the
    compressed fixed / sigma commitments.  (halo2 hashes the Debug string of the pinned verifying key,
    which only the Rust side can produce: the shim passes that value in through zk_pk_set_transcript_repr.)"""
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    h.update(circ.cs_blob())
    for pt in vk_commitments:
        h.update(b.g1_compress(pt))
    return b.fr_from_uniform_bytes(h.digest())


def verify(circ, vk_commitments: Sequence, vk_repr: int, instance: Sequence[Sequence[int]], proof: bytes, s_g2, multiopen: str = "gwc",
           transcript: str = "blake2b") -> bool:
    """halo2_proofs::plonk::verify_proof (KZG, single circuit instance).  vk_commitments: F fixed then P
    sigma affine points (int tuples); s_g2: [s]G2 of the SRS."""
    n, k, u, bf, d = circ.n, circ.k, circ.u, circ.bf, circ.degree()
    F, A, Pn, L = circ.F, circ.A, len(circ.perm_cols), len(circ.lookups)
    chunk = d - 2
    C = (Pn + chunk - 1) // chunk if Pn else 0
    omega = b.omega_for_k(k)
    gates = [circ.compile(g) for g in circ.gates]
    lookups = [([circ.compile(e) for e in lk.table], [[circ.compile(e) for e in i] for i in lk.inputs]) for lk in circ.lookups]
    fixed_com = [Com(pt) for pt in vk_commitments[:F]]
    sigma_com = [Com(pt) for pt in vk_commitments[F:F + Pn]]

    tr = make_transcript(transcript, proof)
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
                adv_com[i] = Com(tr.read_point())
        for ci, cp in enumerate(chal_phase):
            if cp == ph:
                challenges[ci] = tr.squeeze()
    consts = Consts(circ.consts, challenges)
    theta = tr.squeeze()
    m_com = [Com(tr.read_point()) for _ in range(L)]
    beta, gamma = tr.squeeze(), tr.squeeze()
    z_com = [Com(tr.read_point()) for _ in range(C)]
    phi_com = [Com(tr.read_point()) for _ in range(L)]
    random_com = Com(tr.read_point())
    y = tr.squeeze()
    h_com = [tr.read_point() for _ in range(d - 1)]
    x = tr.squeeze()

    rot_last = -(bf + 1)
    point = lambda rot: x * pow(omega, rot % n, R) % R
    ev: Dict[Tuple[int, int, int], int] = {}
    adv_evals = [tr.read_scalar() for _ in circ.advice_queries]
    for (i, rot), e in zip(circ.advice_queries, adv_evals):
        ev[(ADVICE, i, rot)] = e
    fix_evals = [tr.read_scalar() for _ in circ.fixed_queries]
    for (i, rot), e in zip(circ.fixed_queries, fix_evals):
        ev[(FIXED, i, rot)] = e
    random_eval = tr.read_scalar()
    sigma_eval = [tr.read_scalar() for _ in range(Pn)]
    z_eval = []
    for c in range(C):
        e0, e1 = tr.read_scalar(), tr.read_scalar()
        z_eval.append((e0, e1, tr.read_scalar() if c + 1 < C else None))
    lk_eval = [(tr.read_scalar(), tr.read_scalar(), tr.read_scalar()) for _ in range(L)]     # phi(x), phi(wx), m(x)

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

    # ---- expected h(x): gates, permutation argument, lookup arguments, folded with y
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
    for l, (tabs, inputs) in enumerate(lookups):
        # mv_lookup::verifier::Evaluated::expressions
        p0, p1, me = lk_eval[l]
        f_evals = [(compress([eval_program(p, col_eval, consts) for p in ins], theta) + beta) % R for ins in inputs]
        tau = (compress([eval_program(p, col_eval, consts) for p in tabs], theta) + beta) % R
        prod_fi = 1
        for f in f_evals:
            prod_fi = prod_fi * f % R
        if prod_fi == 0 or tau == 0:
            return False                      # batch_invert / invert().unwrap() of the Rust verifier
        sum_inv_fi = sum(b.fr_inv(f) for f in f_evals) % R
        lhs = tau * prod_fi % R * (p1 - p0) % R
        rhs = tau * prod_fi % R * (sum_inv_fi - me * b.fr_inv(tau)) % R
        fold(l0 * p0 % R)
        fold(l_last * p0 % R)
        fold((lhs - rhs) * l_active % R)
    h_eval = acc * b.fr_inv((xn - 1) % R) % R
    # commitment to h(X) = sum_i x^(n i) h_i(X)
    hc = None
    for com in reversed(h_com):
        hc = b.g1_add(b.g1_mul(hc, xn) if hc is not None else None, com)
    h_obj = Com(hc)

    # ---- the multi-open queries, in halo2's order: advice, permutation products, lookups, fixed,
    # permutation sigma, then h and the random polynomial
    queries = [(adv_com[i], point(rot), e) for (i, rot), e in zip(circ.advice_queries, adv_evals)]
    for c in range(C):
        queries += [(z_com[c], point(0), z_eval[c][0]), (z_com[c], point(1), z_eval[c][1])]
    for c in reversed(range(C - 1)):
        queries.append((z_com[c], point(rot_last), z_eval[c][2]))
    for l in range(L):
        queries += [(phi_com[l], point(0), lk_eval[l][0]), (phi_com[l], point(1), lk_eval[l][1]), (m_com[l], point(0), lk_eval[l][2])]
    queries += [(fixed_com[i], point(rot), e) for (i, rot), e in zip(circ.fixed_queries, fix_evals)]
    queries += [(sigma_com[j], point(0), sigma_eval[j]) for j in range(Pn)]
    queries += [(h_obj, point(0), h_eval), (random_com, point(0), random_eval)]
    if multiopen == "shplonk":
        return _verify_shplonk(tr, queries, s_g2)
    return _verify_gwc(tr, queries, s_g2)
