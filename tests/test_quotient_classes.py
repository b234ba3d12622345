"""CPU: the degree-class assembly of constraint programs that share intermediates (csrc/prover.hip TmpSplit, through the
host-only hook zk_host_split_programs).  halo2's GraphEvaluator shares intermediates between gates; the exported programs
carry them as TEE_TMP / PUSH_TMP.  A degree class evaluates only its own constraints, so an intermediate parked by another
class's constraint has to be re-materialised -- every constraint must keep its value whichever class it lands in."""
import ctypes
import random

import numpy as np
import pytest

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
PUSH_COL, PUSH_CONST, ADD, SUB, MUL, NEG, SQUARE, DOUBLE, FOLD, MUL_CONST, ADD_CONST, TEE, PUSH_TMP = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13


def evaluate(prog, cols, consts, tmp, out):
    st = []
    for op, a, b in prog:
        if op == PUSH_COL: st.append(cols[a])
        elif op == PUSH_CONST: st.append(consts[a])
        elif op == ADD: y, x = st.pop(), st.pop(); st.append((x + y) % R)
        elif op == SUB: y, x = st.pop(), st.pop(); st.append((x - y) % R)
        elif op == MUL: y, x = st.pop(), st.pop(); st.append(x * y % R)
        elif op == NEG: st.append(-st.pop() % R)
        elif op == SQUARE: x = st.pop(); st.append(x * x % R)
        elif op == DOUBLE: st.append(2 * st.pop() % R)
        elif op == TEE: tmp[a] = st[-1]
        elif op == PUSH_TMP: st.append(tmp[a])          # KeyError = read before this evaluation defined it
        elif op == FOLD: out[a] = st.pop()
        else: raise AssertionError(op)
    return st


def random_constraints(rng, count, ncols, nconsts, reuse_slots=False):
    """postfix programs; products are parked with TEE_TMP now and then, later constraints pick parked values up"""
    slots = []           # slots defined so far (any constraint)
    progs = []

    def expr(depth, prog):
        pick = rng.random()
        if depth <= 0 or pick < 0.25:
            if slots and rng.random() < 0.4:
                prog.append((PUSH_TMP, rng.choice(slots), 0))
            elif rng.random() < 0.8:
                prog.append((PUSH_COL, rng.randrange(ncols), 0))
            else:
                prog.append((PUSH_CONST, rng.randrange(nconsts), 0))
            return
        if pick < 0.35:
            expr(depth - 1, prog)
            prog.append((rng.choice([NEG, SQUARE, DOUBLE]), 0, 0))
        else:
            expr(depth - 1, prog)
            expr(depth - 1, prog)
            prog.append((rng.choice([ADD, SUB, MUL, MUL]), 0, 0))
        if rng.random() < 0.3:
            s = rng.choice(slots) if (reuse_slots and slots and rng.random() < 0.5) else (max(slots) + 1 if slots else 0)
            prog.append((TEE, s, 0))
            if s not in slots:
                slots.append(s)
    for _ in range(count):
        p = []
        expr(rng.randrange(2, 5), p)
        progs.append(p)
    return progs


def split(zk, progs, cls, classes):
    lib = zk.lib()
    words = np.array([w for p in progs for ins in p for w in ins], dtype=np.uint32)
    lens = np.array([len(p) for p in progs], dtype=np.uint32)
    cls_a = np.array(cls, dtype=np.uint32)
    out_lens = np.zeros(classes, dtype=np.uint32)
    conflict = ctypes.c_int()
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.zk_host_split_programs(ptr(words), ptr(lens), ptr(cls_a), ctypes.c_uint32(len(progs)), ctypes.c_uint32(classes), None, ctypes.c_size_t(0), ptr(out_lens), ctypes.byref(conflict)) == 0
    out = np.zeros(3 * int(out_lens.sum()), dtype=np.uint32)
    assert lib.zk_host_split_programs(ptr(words), ptr(lens), ptr(cls_a), ctypes.c_uint32(len(progs)), ctypes.c_uint32(classes), ptr(out), ctypes.c_size_t(out.size), ptr(out_lens), ctypes.byref(conflict)) == 0
    res, at = [], 0
    for e in range(classes):
        n = int(out_lens[e])
        res.append([tuple(int(x) for x in out[3 * (at + j):3 * (at + j) + 3]) for j in range(n)])
        at += n
    return res, conflict.value


@pytest.mark.parametrize("seed", range(12))
def test_every_constraint_keeps_its_value_in_its_class(zk, seed):
    rng = random.Random(seed)
    ncols, nconsts, count, classes = 6, 3, rng.randrange(4, 14), rng.randrange(1, 5)
    progs = random_constraints(rng, count, ncols, nconsts)
    cls = [rng.randrange(classes) for _ in range(count)]
    class_progs, conflict = split(zk, progs, cls, classes)
    assert conflict == 0
    for _ in range(4):
        cols = [rng.randrange(R) for _ in range(ncols)]
        consts = [rng.randrange(R) for _ in range(nconsts)]
        want, tmp = {}, {}
        for i, p in enumerate(progs):                     # the whole list in order, one shared parking area: what a single class does
            st = evaluate(p, cols, consts, tmp, want)
            assert len(st) == 1
            want[i] = st[0]
        got = {}
        for e in range(classes):                          # every class on its own, starting from an EMPTY parking area
            assert evaluate(class_progs[e], cols, consts, {}, got) == []
        assert got == want
    # a class that holds everything is the original sequence (nothing re-materialised)
    one, _ = split(zk, progs, [0] * count, 1)
    assert [ins for ins in one[0] if ins[0] != FOLD] == [ins for p in progs for ins in p]


def test_slot_reuse_across_constraints_is_reported(zk):
    # slot 0 defined twice, read by a third constraint: which definition a re-materialisation would take is ambiguous -> single class
    progs = [[(PUSH_COL, 0, 0), (PUSH_COL, 1, 0), (MUL, 0, 0), (TEE, 0, 0)],
             [(PUSH_COL, 2, 0), (SQUARE, 0, 0), (TEE, 0, 0)],
             [(PUSH_TMP, 0, 0), (PUSH_COL, 3, 0), (ADD, 0, 0)]]
    _, conflict = split(zk, progs, [0, 1, 0], 2)
    assert conflict == 1
    # the same slot reused but only ever read inside the constraint that wrote it (the prover's lookup identities): fine
    progs = [[(PUSH_COL, 0, 0), (TEE, 5, 0), (PUSH_TMP, 5, 0), (MUL, 0, 0)], [(PUSH_COL, 1, 0), (TEE, 5, 0), (PUSH_TMP, 5, 0), (ADD, 0, 0)]]
    _, conflict = split(zk, progs, [0, 1], 2)
    assert conflict == 0
    # a read before any definition is malformed
    _, conflict = split(zk, [[(PUSH_TMP, 3, 0)]], [0], 1)
    assert conflict == 1


# ------------------------------------------------------------------------------------------------- additive split over classes
def additive_split(zk, prog, E):
    lib = zk.lib()
    words = np.array([w & 0xFFFFFFFF for ins in prog for w in ins], dtype=np.uint32)       # rotations travel as the u32 image of an i32
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    cnt = ctypes.c_uint32()
    assert lib.zk_host_additive_split(ptr(words), ctypes.c_uint32(len(prog)), ctypes.c_uint32(E), None, ctypes.c_size_t(0), None, None, ctypes.c_uint32(0), ctypes.byref(cnt)) == 0
    out = np.zeros(3 * (len(prog) + 8) * max(cnt.value, 1), dtype=np.uint32)
    cls, lens = np.zeros(cnt.value, dtype=np.uint32), np.zeros(cnt.value, dtype=np.uint32)
    assert lib.zk_host_additive_split(ptr(words), ctypes.c_uint32(len(prog)), ctypes.c_uint32(E), ptr(out), ctypes.c_size_t(out.size), ptr(cls), ptr(lens), ctypes.c_uint32(cnt.value), ctypes.byref(cnt)) == 0
    pieces, at = [], 0
    for j in range(cnt.value):
        signed = lambda v: v - (1 << 32) if v >= (1 << 31) else v
        pieces.append((int(cls[j]), [(int(out[3 * (at + i)]), int(out[3 * (at + i) + 1]), signed(int(out[3 * (at + i) + 2]))) for i in range(int(lens[j]))]))
        at += int(lens[j])
    return pieces


def degree(prog):
    st = []
    for op, a, b in prog:
        if op == PUSH_COL: st.append(1)
        elif op == PUSH_CONST: st.append(0)
        elif op in (ADD, SUB): y, x = st.pop(), st.pop(); st.append(max(x, y))
        elif op == MUL: y, x = st.pop(), st.pop(); st.append(x + y)
        elif op == SQUARE: st.append(2 * st.pop())
        elif op in (NEG, DOUBLE, MUL_CONST, ADD_CONST): pass
        else: raise AssertionError(op)
    assert len(st) == 1
    return st[0]


def class_of(deg, E):
    e = 0
    while e < E and (1 << e) < max(deg - 1, 1):
        e += 1
    return e


def test_a_gate_is_split_into_the_classes_of_its_terms(zk):
    """q (a b - c): the product is of degree 3 (two cosets), q c of degree 2 (one coset) -- the shape of the multiplication gates of
    the SuperCircuit stand-in (bench_proof.build_shape); the output column c is then read on one coset only"""
    E = 3
    q, a, b_, c = (PUSH_COL, 0, 0), (PUSH_COL, 1, 0), (PUSH_COL, 2, 0), (PUSH_COL, 3, 0)
    gate = [q, a, b_, (MUL, 0, 0), c, (SUB, 0, 0), (MUL, 0, 0)]
    pieces = additive_split(zk, gate, E)
    assert sorted(cl for cl, _ in pieces) == [0, 1]
    by = dict(pieces)
    assert 3 not in [ins[1] for ins in by[1] if ins[0] == PUSH_COL]         # the class of the product does not read c
    assert [ins[1] for ins in by[0] if ins[0] == PUSH_COL] == [0, 3]         # the low class reads q and c only
    # the selector on the other side, a sum without a factor, a linear gate (not split: one class), a product (not a sum: whole)
    assert sorted(cl for cl, _ in additive_split(zk, [a, b_, (MUL, 0, 0), c, (SUB, 0, 0), q, (MUL, 0, 0)], E)) == [0, 1]
    assert sorted(cl for cl, _ in additive_split(zk, [a, b_, (MUL, 0, 0), a, (MUL, 0, 0), a, (MUL, 0, 0), c, (ADD, 0, 0)], E)) == [0, 2]
    assert [cl for cl, _ in additive_split(zk, [q, a, b_, (ADD, 0, 0), c, (SUB, 0, 0), (MUL, 0, 0)], E)] == [0]
    assert [cl for cl, _ in additive_split(zk, [q, a, (MUL, 0, 0), b_, (MUL, 0, 0), c, (MUL, 0, 0)], E)] == [2]
    # parked intermediates: left whole
    assert len(additive_split(zk, [q, a, b_, (MUL, 0, 0), (TEE, 0, 0), c, (SUB, 0, 0), (MUL, 0, 0)], E)) == 1


@pytest.mark.parametrize("seed", range(16))
def test_the_pieces_of_a_constraint_sum_to_it(zk, seed):
    rng = random.Random(1000 + seed)
    ncols, nconsts, E = 6, 3, rng.randrange(1, 4)

    def term(depth, prog):
        if depth <= 0 or rng.random() < 0.3:
            prog.append((PUSH_COL, rng.randrange(ncols), rng.randrange(-1, 2)) if rng.random() < 0.8 else (PUSH_CONST, rng.randrange(nconsts), 0))
            return
        pick = rng.random()
        if pick < 0.2:
            term(depth - 1, prog)
            prog.append((rng.choice([NEG, SQUARE, DOUBLE, MUL_CONST, ADD_CONST]), rng.randrange(nconsts), 0))
        else:
            term(depth - 1, prog)
            term(depth - 1, prog)
            prog.append((rng.choice([ADD, SUB, MUL, MUL, MUL]), 0, 0))

    def build_sum(count, prog):
        term(rng.randrange(0, 3), prog)
        if rng.random() < 0.3:
            prog.append((NEG, 0, 0))
        for _ in range(count - 1):
            term(rng.randrange(0, 3), prog)
            prog.append((rng.choice([ADD, SUB]), 0, 0))

    for _ in range(20):
        prog = []
        shape = rng.randrange(3)
        if shape == 0:
            build_sum(rng.randrange(2, 6), prog)
        elif shape == 1:
            prog.append((PUSH_COL, 0, 0))
            build_sum(rng.randrange(2, 6), prog)
            prog.append((MUL, 0, 0))
        else:
            build_sum(rng.randrange(2, 6), prog)
            prog.append((PUSH_COL, 0, 0))
            prog.append((MUL, 0, 0))
        if degree(prog) > (1 << E) + 1:
            continue
        pieces = additive_split(zk, prog, E)
        classes = [cl for cl, _ in pieces]
        assert classes == sorted(set(classes))                    # ascending, one piece per class
        assert max(classes) == class_of(degree(prog), E)            # the highest class is the constraint's own
        for cl, pg in pieces:
            assert class_of(degree(pg), E) == cl or len(pieces) == 1

        def run(pg, cols, consts):
            st = []
            for op, a, b in pg:
                if op == PUSH_COL: st.append(cols[(a, b)])
                elif op == PUSH_CONST: st.append(consts[a])
                elif op == ADD: y, x = st.pop(), st.pop(); st.append((x + y) % R)
                elif op == SUB: y, x = st.pop(), st.pop(); st.append((x - y) % R)
                elif op == MUL: y, x = st.pop(), st.pop(); st.append(x * y % R)
                elif op == NEG: st.append(-st.pop() % R)
                elif op == SQUARE: x = st.pop(); st.append(x * x % R)
                elif op == DOUBLE: st.append(2 * st.pop() % R)
                elif op == MUL_CONST: st.append(st.pop() * consts[a] % R)
                elif op == ADD_CONST: st.append((st.pop() + consts[a]) % R)
                else: raise AssertionError(op)
            assert len(st) == 1
            return st[0]
        for _ in range(3):
            cols = {(c_, r_): rng.randrange(R) for c_ in range(ncols) for r_ in (-1, 0, 1)}
            consts = [rng.randrange(R) for _ in range(nconsts)]
            assert sum(run(pg, cols, consts) for _, pg in pieces) % R == run(prog, cols, consts)


# ------------------------------------------------------------------------------------------------- one weighted sum per class
YPOW0 = 0xFFFC0000


def group_terms(zk, terms, cons, K):
    lib = zk.lib()
    words = np.array([w & 0xFFFFFFFF for p in terms for ins in p for w in ins], dtype=np.uint32)
    lens = np.array([len(p) for p in terms], dtype=np.uint32)
    cons_a = np.array(cons, dtype=np.uint32)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    cnt = ctypes.c_uint32()
    rc = lib.zk_host_group_terms(ptr(words), ptr(lens), ptr(cons_a), ctypes.c_uint32(len(terms)), ctypes.c_uint32(K), None, ctypes.c_size_t(0), ctypes.byref(cnt))
    if rc != 0:
        return None
    out = np.zeros(3 * cnt.value, dtype=np.uint32)
    assert lib.zk_host_group_terms(ptr(words), ptr(lens), ptr(cons_a), ctypes.c_uint32(len(terms)), ctypes.c_uint32(K), ptr(out), ctypes.c_size_t(out.size), ctypes.byref(cnt)) == 0
    return [tuple(int(x) for x in out[3 * j:3 * j + 3]) for j in range(cnt.value)]


def evaluate_weighted(prog, cols, consts, y, tmp):
    """the evaluator of the grouped form: MUL_CONST by a power of y (0xFFFC0000 + g) or by a circuit constant, one closing FOLD"""
    st = []
    for op, a, b in prog:
        if op == PUSH_COL: st.append(cols[a])
        elif op == PUSH_CONST: st.append(consts[a])
        elif op == PUSH_TMP: st.append(tmp[a])
        elif op == TEE: tmp[a] = st[-1]
        elif op == ADD: y_, x_ = st.pop(), st.pop(); st.append((x_ + y_) % R)
        elif op == SUB: y_, x_ = st.pop(), st.pop(); st.append((x_ - y_) % R)
        elif op == MUL: y_, x_ = st.pop(), st.pop(); st.append(x_ * y_ % R)
        elif op == NEG: st.append(-st.pop() % R)
        elif op == SQUARE: x_ = st.pop(); st.append(x_ * x_ % R)
        elif op == DOUBLE: st.append(2 * st.pop() % R)
        elif op == MUL_CONST: st.append(st.pop() * (pow(y, a - YPOW0, R) if a >= YPOW0 else consts[a]) % R)
        elif op == ADD_CONST: st.append((st.pop() + consts[a]) % R)
        elif op == FOLD: assert len(st) == 1; return st.pop()
        else: raise AssertionError(op)
    raise AssertionError("no closing FOLD")


def weighted_reference(terms, cons, K, cols, consts, y):
    tmp, total = {}, 0
    for p, i in zip(terms, cons):
        st = evaluate(p_plain(p, consts), cols, consts, tmp, {})
        assert len(st) == 1
        total = (total + pow(y, K - 1 - i, R) * st[0]) % R
    return total


def p_plain(p, consts):
    """MUL_CONST / ADD_CONST rewritten as PUSH_CONST + MUL / ADD for the plain evaluator"""
    out = []
    for op, a, b in p:
        if op == MUL_CONST: out += [(PUSH_CONST, a, 0), (MUL, 0, 0)]
        elif op == ADD_CONST: out += [(PUSH_CONST, a, 0), (ADD, 0, 0)]
        else: out.append((op, a, b))
    return out


def test_terms_under_a_shared_selector_are_collected(zk):
    """the terms of the SuperCircuit stand-in's gate classes: q_s (a b) and q_s c under a handful of selectors, a term nobody shares a
    factor with, and the remainder columns at weight one"""
    rng = random.Random(5)
    S, G = 3, 11
    col = lambda i: (PUSH_COL, i, 0)
    terms, cons = [], []
    for g in range(G):
        q, a, b_ = col(g % S), col(S + 2 * g), col(S + 2 * g + 1)
        terms.append([q, a, b_, (MUL, 0, 0), (MUL, 0, 0)] if g % 2 else [a, b_, (MUL, 0, 0), q, (MUL, 0, 0)])     # factor first / factor last
        cons.append(2 * g)
    lone = [col(S), (SQUARE, 0, 0), col(S + 1), (ADD, 0, 0), (MUL_CONST, 1, 0)]
    terms.append(lone); cons.append(2 * G)
    terms.append([col(S + 3), (NEG, 0, 0)]); cons.append(2 * G + 4)           # a remainder: weight y^0 when it is the last constraint
    K = 2 * G + 5
    prog = group_terms(zk, terms, cons, K)
    assert prog is not None
    # one product per selector, none for the folding: G products a*b, G + 1 weights (the lone term's too), S selector products
    assert sum(1 for ins in prog if ins[0] == MUL) == G + S
    assert sum(1 for ins in prog if ins[0] == MUL_CONST and ins[1] >= YPOW0) == G + 1
    assert sum(1 for ins in prog if ins[0] == FOLD) == 1 and prog[-1][0] == FOLD
    ncols = S + 2 * G + 2
    for _ in range(4):
        cols = [rng.randrange(R) for _ in range(ncols)]
        consts = [rng.randrange(R) for _ in range(3)]
        y = rng.randrange(R)
        assert evaluate_weighted(prog, cols, consts, y, {}) == weighted_reference(terms, cons, K, cols, consts, y)


def test_parked_intermediates_stay_ahead_of_their_readers(zk):
    """the lookup identities: l_active * (phi_f ((phi' - phi) tau + m) - tau) with tau parked by the first lookup into a table and read
    back by the others; a second table parks its own; a term that re-parks a slot it alone reads (the multi-tuple form) is left where it
    stands"""
    rng = random.Random(9)
    col = lambda i, rot=0: (PUSH_COL, i, rot)
    lact, ta, tb = col(0), col(1), col(2)
    terms, cons = [], []
    nl = 7
    for l in range(nl):
        table = l % 2                               # two tables, slots 4 and 5
        slot = 4 + table
        first = l < 2
        tau = [ta if table == 0 else tb, (ADD_CONST, 0, 0), (TEE, slot, 0)] if first else [(PUSH_TMP, slot, 0)]
        phi, m, f = 3 + 3 * l, 4 + 3 * l, 5 + 3 * l
        t = [lact, col(phi, 1), col(phi), (SUB, 0, 0)] + tau + [(MUL, 0, 0), col(m), (ADD, 0, 0), col(f), (ADD_CONST, 0, 0), (MUL, 0, 0), (PUSH_TMP, slot, 0), (SUB, 0, 0), (MUL, 0, 0)]
        terms.append(t); cons.append(3 * l + 2)
    # a multi-tuple style term in between: parks slot 0 twice over the list, reads it itself
    for pos in (2, 5):
        t = [lact, col(3 + pos), (ADD_CONST, 1, 0), (TEE, 0, 0), (PUSH_TMP, 0, 0), (MUL, 0, 0), (MUL, 0, 0)]
        terms.insert(pos, t); cons.insert(pos, cons[pos - 1] + 1)
    K = max(cons) + 2
    prog = group_terms(zk, terms, cons, K)
    assert prog is not None
    # every read of a slot comes after a definition of it
    seen = set()
    for op, a, b in prog:
        if op == TEE: seen.add(a)
        if op == PUSH_TMP: assert a in seen
    ncols = 3 + 3 * nl + 3
    for _ in range(4):
        cols = [rng.randrange(R) for _ in range(ncols)]
        consts = [rng.randrange(R) for _ in range(2)]
        y = rng.randrange(R)
        assert evaluate_weighted(prog, cols, consts, y, {}) == weighted_reference(terms, cons, K, cols, consts, y)
    # the seven single-tuple lookups share ONE product by l_active
    assert sum(1 for i, ins in enumerate(prog) if ins[0] == MUL and prog[i - 1] == lact) <= 3


@pytest.mark.parametrize("seed", range(10))
def test_random_terms_keep_their_weighted_sum(zk, seed):
    rng = random.Random(100 + seed)
    ncols, nconsts, count = 5, 3, rng.randrange(3, 16)
    progs = random_constraints(rng, count, ncols, nconsts, reuse_slots=(seed % 3 == 0))
    # give some of them a shared single-column factor, in front or behind
    terms = []
    for p in progs:
        r = rng.random()
        f = (PUSH_COL, rng.randrange(2), 0)
        terms.append([f] + p + [(MUL, 0, 0)] if r < 0.35 else (p + [f, (MUL, 0, 0)] if r < 0.7 else p))
    cons = sorted(rng.sample(range(3 * count), count))
    K = 3 * count + 1
    prog = group_terms(zk, terms, cons, K)
    if prog is None:
        pytest.skip("kept folded (stack depth)")
    for _ in range(3):
        cols = [rng.randrange(R) for _ in range(ncols)]
        consts = [rng.randrange(R) for _ in range(nconsts)]
        y = rng.randrange(R)
        try:
            want = weighted_reference(terms, cons, K, cols, consts, y)
        except KeyError:
            pytest.skip("a term reads a slot nobody parked before it (malformed input)")
        assert evaluate_weighted(prog, cols, consts, y, {}) == want
