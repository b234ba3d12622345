"""CPU: the class-program compiler of round 6 (csrc/class_compile.hpp, through the host-only hooks zk_host_compile_class and
zk_host_quotient_plan): hash-consed expression graph, regrouping under common factors of any shape, Horner sums, parking slots by
liveness.  Whatever it does to a class's terms, the program must leave  acc = sum_t y^(last - cons_t) term_t  -- checked against
big-int evaluation of the ORIGINAL terms -- and what the kernel then runs (zk_host_quotient_lower) must keep every bound of the
29-bit-limb arithmetic with slots that are reused (the limb-level executor of test_quotient_lowering).  The EVM-style constraint
system of bench_proof.evm_block [REF zkevm-circuits/src/evm_circuit/execution.rs:832-851] is the workload the pass exists for."""
import ctypes
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from test_quotient_classes import (ADD, ADD_CONST, DOUBLE, FOLD, MUL, MUL_CONST, NEG, PUSH_COL, PUSH_CONST, PUSH_TMP, R, SQUARE, SUB, TEE, evaluate,  # noqa: E402
                                   p_plain, random_constraints)

YPOW0, C_ONE = 0xFFFC0000, 0xFFFF0004


def compile_class(zk, terms, cons, K):
    lib = zk.lib()
    words = np.array([w & 0xFFFFFFFF for p in terms for ins in p for w in ins], dtype=np.uint32)
    lens = np.array([len(p) for p in terms], dtype=np.uint32)
    cons_a = np.array(cons, dtype=np.uint32)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    cnt, last = ctypes.c_uint32(), ctypes.c_uint32()
    stats = np.zeros(6, dtype=np.uint32)
    rc = lib.zk_host_compile_class(ptr(words), ptr(lens), ptr(cons_a), ctypes.c_uint32(len(terms)), ctypes.c_uint32(K), None, ctypes.c_size_t(0), ctypes.byref(cnt), ctypes.byref(last), ptr(stats))
    if rc != 0:
        return None
    out = np.zeros(3 * cnt.value, dtype=np.uint32)
    assert lib.zk_host_compile_class(ptr(words), ptr(lens), ptr(cons_a), ctypes.c_uint32(len(terms)), ctypes.c_uint32(K), ptr(out), ctypes.c_size_t(out.size), ctypes.byref(cnt), ctypes.byref(last), ptr(stats)) == 0
    prog = [tuple(int(x) for x in out[3 * j:3 * j + 3]) for j in range(cnt.value)]
    return prog, last.value, dict(zip(("nodes", "parked", "max_live", "products", "groups", "depth"), (int(v) for v in stats)))


def run_compiled(prog, cols, consts, y):
    """the accumulator machine: FOLD c = (acc = acc * c + pop); constants above YPOW0 are powers of y, C_ONE is 1"""
    def cst(a):
        return 1 if a == C_ONE else (pow(y, a - YPOW0, R) if a >= YPOW0 else consts[a])
    st, tmp, acc = [], {}, 0
    for op, a, b in prog:
        if op == PUSH_COL: st.append(cols[(a, b)] if isinstance(cols, dict) else cols[a])
        elif op == PUSH_CONST: st.append(cst(a))
        elif op == PUSH_TMP: st.append(tmp[a])
        elif op == TEE: tmp[a] = st[-1]
        elif op == ADD: y_, x_ = st.pop(), st.pop(); st.append((x_ + y_) % R)
        elif op == SUB: y_, x_ = st.pop(), st.pop(); st.append((x_ - y_) % R)
        elif op == MUL: y_, x_ = st.pop(), st.pop(); st.append(x_ * y_ % R)
        elif op == NEG: st.append(-st.pop() % R)
        elif op == SQUARE: x_ = st.pop(); st.append(x_ * x_ % R)
        elif op == DOUBLE: st.append(2 * st.pop() % R)
        elif op == MUL_CONST: st.append(st.pop() * cst(a) % R)
        elif op == ADD_CONST: st.append((st.pop() + cst(a)) % R)
        elif op == FOLD: acc = (acc * cst(a) + st.pop()) % R
        else: raise AssertionError(op)
    assert not st
    return acc


def reference(terms, cons, last, cols, consts, y):
    tmp, total = {}, 0
    for p, i in zip(terms, cons):
        st = evaluate(p_plain(p, consts), cols, consts, tmp, {})
        assert len(st) == 1
        total = (total + pow(y, last - i, R) * st[0]) % R
    return total


@pytest.mark.parametrize("seed", range(16))
def test_random_terms_keep_their_weighted_sum(zk, seed):
    rng = random.Random(300 + seed)
    ncols, nconsts, count = 5, 3, rng.randrange(3, 24)
    progs = random_constraints(rng, count, ncols, nconsts, reuse_slots=(seed % 3 == 0))
    terms = []
    for p in progs:                      # shared factors of several shapes: a column, a product of columns, in front or behind
        r = rng.random()
        f = [(PUSH_COL, rng.randrange(2), 0)] if rng.random() < 0.5 else [(PUSH_COL, 0, 0), (PUSH_COL, 1 + rng.randrange(2), 0), (MUL, 0, 0)]
        terms.append(f + p + [(MUL, 0, 0)] if r < 0.35 else (p + f + [(MUL, 0, 0)] if r < 0.7 else p))
    cons = sorted(rng.choices(range(3 * count), k=count))           # several terms may belong to one constraint (the pieces of a split)
    K = 3 * count + 1
    got = compile_class(zk, terms, cons, K)
    if got is None:
        pytest.skip("kept in the old form (stack depth)")
    prog, last, st = got
    assert last == max(cons)
    seen = set()
    for op, a, b in prog:                # every read of a slot comes after a write of it
        if op == TEE: seen.add(a)
        if op == PUSH_TMP: assert a in seen
    for _ in range(3):
        cols = [rng.randrange(R) for _ in range(ncols)]
        consts = [rng.randrange(R) for _ in range(nconsts)]
        y = rng.randrange(R)
        try:
            want = reference(terms, cons, last, cols, consts, y)
        except KeyError:
            pytest.skip("a term reads a slot nobody parked before it (malformed input)")
        assert run_compiled(prog, cols, consts, y) == want


def evm_terms(states=6, per_state=16, input_cols=6, cond_cols=2, k=7):
    """the gates of bench_proof.evm_block as postfix programs over abstract column references (what zk_pk_create reads)"""
    import bench_proof as bp
    from zkevm_circuits_amd import plonk
    p = {"states": states, "per_state": per_state, "cond_cols": cond_cols, "input_cols": input_cols, "seed": 5}
    c = plonk.Circuit(k, num_fixed=1, num_advice=bp.evm_step_columns(p), num_instance=0, blinding_factors=5)
    spec = bp.evm_block(c, 0, c.fixed_col(0), p)
    terms = [[(op, a, b if b < (1 << 31) else b - (1 << 32)) for op, a, b in c.compile(g)] for g in c.gates]
    return c, spec, terms


def test_evm_style_constraints_share_their_selector_products(zk):
    """q_usable * q_step * state_selector_s multiplies ONCE per state, a gadget's condition once per gadget; the values alive in the
    parking area at once are a handful although hundreds of sub-expressions are shared"""
    states, per_state = 6, 16
    c, spec, terms = evm_terms(states, per_state)
    cons = list(range(len(terms)))
    K = len(terms)
    prog, last, st = compile_class(zk, terms, cons, K)
    assert last == K - 1
    plain_products = sum(1 for p in terms for ins in p if ins[0] == MUL) + K           # as exported, folded one by one
    assert st["products"] < 0.5 * plain_products, (st, plain_products)
    assert st["groups"] >= states                       # at least one factor group per execution state
    assert st["max_live"] <= 8 and st["depth"] <= 14
    # q_usable * q_step: once per state at most (ONE product, alive over the whole program: recomputed rather than parked), not once per constraint
    q_usable, q_step = (PUSH_COL, 0 << 24 | 0, 0), (PUSH_COL, 1 << 24 | spec["q_step"], 0)
    assert sum(1 for i in range(len(prog) - 2) if set(prog[i:i + 2]) == {q_usable, q_step} and prog[i + 2][0] == MUL) <= states + 1
    rng = random.Random(1)
    consts = [c_ % R for c_ in c.consts]
    for _ in range(2):
        cols = {}
        for p in terms:
            for op, a, b in p:
                if op == PUSH_COL: cols.setdefault((a, b), rng.randrange(R))
        y = rng.randrange(R)
        want = sum(pow(y, last - i, R) * evaluate_rot(p, cols, consts) for p, i in zip(terms, cons)) % R
        assert run_compiled(prog, cols, consts, y) == want


@pytest.mark.parametrize("chunk", ["1", "0"])
def test_chunked_emission_keeps_the_value_and_frees_every_fold_boundary(zk, chunk, monkeypatch):
    """Round 6, second half: large class programs are emitted term by term (ClassCompiler::chunked, forced here by ZK_QUOTIENT_CHUNK=1 on a small one) so that
    no parked value is alive across a top-level FOLD -- the evaluator may then cut the program into slices there (tests/test_quotient_slices.py).  The value
    stays the weighted sum of the terms; with one parking scope for the whole class (ZK_QUOTIENT_CHUNK=0) some boundaries are crossed by parked values."""
    monkeypatch.setenv("ZK_QUOTIENT_CHUNK", chunk)
    c, spec, terms = evm_terms(states=6, per_state=16)
    cons = list(range(len(terms)))
    K = len(terms)
    prog, last, st = compile_class(zk, terms, cons, K)
    rng = random.Random(11)
    consts = [c_ % R for c_ in c.consts]
    cols = {}
    for p in terms:
        for op, a, b in p:
            if op == PUSH_COL: cols.setdefault((a, b), rng.randrange(R))
    y = rng.randrange(R)
    want = sum(pow(y, last - i, R) * evaluate_rot(p, cols, consts) for p, i in zip(terms, cons)) % R
    assert run_compiled(prog, cols, consts, y) == want
    # top-level folds and the parked values alive across them
    sp, alive_from, crossed, folds = 0, {}, 0, 0
    last_read = {}
    for pc, (op, a, b) in enumerate(prog):
        if op == PUSH_TMP: last_read[(a, alive_from[a])] = pc
        if op == TEE: alive_from[a] = pc
    spans = [(t, u) for (slot, t), u in last_read.items()]
    for pc, (op, a, b) in enumerate(prog):
        if op in (PUSH_COL, PUSH_CONST, PUSH_TMP): sp += 1
        elif op in (ADD, SUB, MUL): sp -= 1
        elif op == FOLD:
            sp -= 1
            if sp == 0:
                folds += 1
                crossed += any(t < pc + 1 <= u for t, u in spans)
    assert folds >= 6
    if chunk == "1":
        assert crossed == 0
    else:
        assert crossed > 0            # what made the compiled EVM-style program unsliceable before


def evaluate_rot(prog, cols, consts):
    st = []
    for op, a, b in prog:
        if op == PUSH_COL: st.append(cols[(a, b)])
        elif op == PUSH_CONST: st.append(consts[a])
        elif op == ADD: y, x = st.pop(), st.pop(); st.append((x + y) % R)
        elif op == SUB: y, x = st.pop(), st.pop(); st.append((x - y) % R)
        elif op == MUL: y, x = st.pop(), st.pop(); st.append(x * y % R)
        elif op == NEG: st.append(-st.pop() % R)
        else: raise AssertionError(op)
    assert len(st) == 1
    return st[0]


def test_evm_style_program_keeps_the_limb_bounds_when_lowered(zk):
    """compile -> lower (what the kernel runs, with reused slots) -> executed limb by limb at adversarial operand values"""
    import test_quotient_lowering as tl
    c, spec, terms = evm_terms(states=4, per_state=8, input_cols=4, cond_cols=2)
    K = len(terms)
    prog, last, st = compile_class(zk, terms, list(range(K)), K)
    # concretise: columns (ref, rot) -> indices, abstract constants -> a table
    col_ix, const_ix, consts_tab = {}, {}, []
    rng = random.Random(3)
    y = rng.randrange(R)

    def cst(a):
        if a not in const_ix:
            v = 1 if a == C_ONE else (pow(y, a - YPOW0, R) if a >= YPOW0 else c.consts[a] % R)
            const_ix[a] = len(consts_tab)
            consts_tab.append(v)
        return const_ix[a]
    conc = []
    for op, a, b in prog:
        if op == PUSH_COL:
            col_ix.setdefault((a, b), len(col_ix))
            conc.append((op, col_ix[(a, b)], 0))
        elif op in (PUSH_CONST, MUL_CONST, ADD_CONST, FOLD):
            conc.append((op, cst(a), 0))
        else:
            conc.append((op, a, b))
    ncols = len(col_ix)
    words, depth = tl.lower(conc, ncols)
    assert depth <= 16
    RR = 1 << 256
    for trial in range(3):
        pick = [lambda: 0, lambda: 1, lambda: R - 1, lambda: rng.randrange(R)]
        vals = {key: (pick[rng.randrange(4)]() if trial else rng.randrange(R)) for key in col_ix}      # canonical R-form integers
        lowered_cols = {(i, 0): vals[key] for key, i in col_ix.items()}
        got = tl.run_lowered(words, lowered_cols, [v * RR % R for v in consts_tab], ncols)
        # plain: values are x R, products divide by R -> compare through the plain integer evaluation of the original terms
        rinv = pow(RR, -1, R)
        cols_plain = {key: v * rinv % R for key, v in vals.items()}
        want = 0
        for i, p in enumerate(terms):
            want = (want + pow(y, last - i, R) * evaluate_rot(p, cols_plain, [x % R for x in c.consts])) % R
        assert got == want * RR % R


def test_slots_are_reused_by_liveness(zk):
    """forty expensive sub-expressions, each shared by two neighbouring terms: forty values parked, one or two alive at a time"""
    col = lambda i: (PUSH_COL, i, 0)
    terms = []
    for g in range(40):
        shared = [col(3 * g), col(3 * g + 1), (MUL, 0, 0), col(3 * g + 2), (MUL, 0, 0)]            # two products: worth parking
        terms.append(shared + [col(200), (ADD, 0, 0)])
        terms.append(shared + [col(201), (SUB, 0, 0)])
    K = len(terms)
    prog, last, st = compile_class(zk, terms, list(range(K)), K)
    assert st["parked"] == 40 and st["max_live"] <= 2
    assert max(a for op, a, b in prog if op == TEE) <= 1
    rng = random.Random(4)
    cols = [rng.randrange(R) for _ in range(202)]
    y = rng.randrange(R)
    assert run_compiled(prog, cols, [], y) == reference(terms, list(range(K)), last, cols, [], y)


def test_cheap_values_are_not_parked_across_long_spans(zk):
    """cell * 256 read again thousands of instructions later is recomputed, not held in the parking area"""
    col = lambda i: (PUSH_COL, i, 0)
    terms = [[col(0), (PUSH_CONST, 0, 0), (MUL, 0, 0), col(1), (ADD, 0, 0)]]
    for g in range(300):
        terms.append([col(2 + g), col(3 + g), (MUL, 0, 0), col(4 + g), (SUB, 0, 0)])
    terms.append([col(0), (PUSH_CONST, 0, 0), (MUL, 0, 0), col(5), (SUB, 0, 0)])
    K = len(terms)
    prog, last, st = compile_class(zk, terms, list(range(K)), K)
    assert st["parked"] == 0
    rng = random.Random(6)
    cols = [rng.randrange(R) for _ in range(400)]
    y = rng.randrange(R)
    assert run_compiled(prog, cols, [256], y) == reference(terms, list(range(K)), last, cols, [256], y)


def test_the_plan_of_an_evm_style_constraint_system(zk):
    """zk_host_quotient_plan on the cs part of the key blob alone: >= 5 000 constraints of degree 5..9 compile to a program a third the
    size of the exported trees, with a parking area of a few slots; every knob combination yields a plan"""
    import bench_proof as bp
    from zkevm_circuits_amd import plonk
    p = dict(bp.EVM_DEFAULT)
    S = bp.evm_step_columns(p)
    assert 150 <= S <= 165
    c = plonk.Circuit(10, num_fixed=1, num_advice=S, num_instance=0, blinding_factors=5)
    bp.evm_block(c, 0, c.fixed_col(0), p)
    degs = [g.degree() for g in c.gates]
    assert sum(1 for d in degs if 5 <= d <= 9) >= 5000 and max(degs) == 9
    lib = zk.lib()
    blob = c.cs_blob()
    E = c.extended_k() - c.k

    def plan():
        summ = np.zeros(8 + 8 * (E + 1), dtype=np.uint32)
        n_ = ctypes.c_uint32()
        assert lib.zk_host_quotient_plan(blob, ctypes.c_size_t(len(blob)), summ.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(summ.size), ctypes.c_uint32(E), None, ctypes.c_size_t(0), ctypes.byref(n_)) == 0
        return summ
    s = plan()
    assert s[0] == E and s[1] == len(c.gates) and s[4] == 1
    instr = sum(int(s[8 + 8 * e + 1]) for e in range(E + 1))
    products = sum(int(s[8 + 8 * e + 2]) for e in range(E + 1))
    live = max(int(s[8 + 8 * e + 5]) for e in range(E + 1))
    assert instr >= 50000 and live <= 64
    os.environ["ZK_QUOTIENT_DAG"] = "0"
    try:
        s0 = plan()
    finally:
        os.environ.pop("ZK_QUOTIENT_DAG")
    products0 = sum(int(s0[8 + 8 * e + 2]) for e in range(E + 1))
    assert s0[4] == 0 and products < 0.4 * products0, (products, products0)
    for env in ({"ZK_QUOTIENT_SPLIT": "0"}, {"ZK_QUOTIENT_ADDSPLIT": "0"}, {"ZK_QUOTIENT_COSTGATE": "0"}, {"ZK_QUOTIENT_GROUP": "0", "ZK_QUOTIENT_DAG": "0"}):
        os.environ.update(env)
        try:
            sx = plan()
            if "ZK_QUOTIENT_SPLIT" in env:
                assert sx[2] == 0 and sx[8 + 8 * E] == 1 and all(sx[8 + 8 * e] == 0 for e in range(E))
            if "ZK_QUOTIENT_COSTGATE" in env:
                assert sx[2] == 1 and sx[3] == 1
        finally:
            for k_ in env:
                os.environ.pop(k_)
