"""The expression evaluator's lowering pass (csrc/quotient.hip: lower_fuse, lower_bounds), checked on the CPU.

`zk_host_quotient_lower` returns the instruction stream the kernel runs for a caller's postfix program.  This test
executes that stream the way the kernel does -- nine 29-bit limbs in 32-bit words, 64-bit column sums, unsettled sums --
with every precondition of ff29.hip.hpp asserted (no 32-bit limb overflow, no 64-bit column overflow, no borrow out of a
top limb that a product would read, settle only below 4p, ...), and compares the canonical result with plain big-int
evaluation of the ORIGINAL program.  Columns hold adversarial values (0, 1, p - 1, random), so the static bounds the
lowering derives are exercised at their edges.  No GPU involved: the kernel's arithmetic itself is covered by the
`-m gpu` tests (test_gpu_quotient.py, proof byte parity).
"""
import ctypes
import random

import numpy as np
import pytest

from zkevm_circuits_amd import binding

P = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
MASK = (1 << 29) - 1
M = [(P >> (29 * i)) & MASK if i < 8 else P >> 232 for i in range(9)]
INV = 0xfffffff
R = 1 << 256

(Q_END, Q_PUSH_COL, Q_PUSH_CONST, Q_ADD, Q_SUB, Q_MUL, Q_NEG, Q_SQUARE, Q_DOUBLE, Q_FOLD, Q_MUL_CONST, Q_ADD_CONST, Q_TEE_TMP,
 Q_PUSH_TMP) = range(14)
K_ADD_COL, K_SUB_COL, K_RSUB_COL, K_MUL_COL, K_FOLD_COL, K_NOP = 16, 17, 18, 19, 20, 21


# ---- limb-level model of ff29.hip.hpp ------------------------------------------------------------------
def val(l):
    return sum(x << (29 * i) for i, x in enumerate(l))


def unpack(x, bits=256):
    assert 0 <= x < (1 << bits)
    return [(x >> (29 * i)) & MASK for i in range(8)] + [x >> 232]


def add29(a, b):
    r = [x + y for x, y in zip(a, b)]
    assert all(x < (1 << 32) for x in r), "limb overflow in add29"
    return r


def kp_balanced(K, idx):
    n = unpack(K * P, 261)
    out = n[idx]
    if idx < 8:
        out += 1 << 29
    if idx > 0:
        out -= 1
    return out


def sub29k(K, a, b, *, normalised_after):
    assert all(x <= MASK for x in b[:8]), "subtrahend must be normalised"
    assert val(b) < K * P, "subtrahend must be below K p"
    r = []
    for i in range(9):
        v = a[i] + kp_balanced(K, i) - b[i]
        if i < 8:
            assert 0 <= v < (1 << 32), "limb under/overflow in sub29k"
        else:
            # the top limb may borrow (it is only correct modulo 2^32 until carries are propagated into it)
            assert normalised_after or v >= 0, "negative top limb would be read by a product"
            assert v < (1 << 32)
            v &= 0xffffffff
        r.append(v)
    return r


def normalize29(a):
    a = list(a)
    for i in range(8):
        a[i + 1] = (a[i + 1] + (a[i] >> 29)) & 0xffffffff
        a[i] &= MASK
    return a


def settle(a):
    assert all(x < (1 << 31) for x in a), "q_settle: limbs must be below 2^31"
    a = normalize29(a)
    v = val(a)
    assert v < 4 * P, "q_settle: value must be below 4p"
    if v >= 2 * P:
        v -= 2 * P
    return unpack(v)


def settle8(a):
    assert all(x < (1 << 31) for x in a), "q_settle8: limbs must be below 2^31"
    a = normalize29(a)
    v = val(a)
    assert v < 8 * P, "q_settle8: value must be below 8p"
    if v >= 4 * P:
        v -= 4 * P
    return settle(unpack(v, 261))


def mul29(a, b):
    assert all(x <= MASK + 8 for x in b[:8]), "second operand of mul29 must be normalised"
    assert val(a) * val(b) < (1 << 261) * P, "mul29: a b must be below 2^261 p"
    m = [0] * 9
    t = [0] * 9
    acc = 0
    for k in range(9):
        for i in range(k + 1):
            acc += a[i] * b[k - i]
        for i in range(k):
            acc += m[i] * M[k - i]
        assert acc < (1 << 64), "column overflow"
        m[k] = ((acc & 0xffffffff) * INV) & MASK
        acc += m[k] * M[0]
        assert acc < (1 << 64), "column overflow"
        acc >>= 29
    for k in range(9, 17):
        for i in range(k - 8, 9):
            acc += a[i] * b[k - i] + m[i] * M[k - i]
        assert acc < (1 << 64), "column overflow"
        t[k - 9] = acc & MASK
        acc >>= 29
    t[8] = acc
    assert acc < (1 << 32)
    assert val(t) < 2 * P and (val(t) << 261) % P == (val(a) * val(b)) % P
    return t


def shl5(x):
    assert all(v <= MASK for v in x[:8]) and val(x) < 2 * P
    return unpack(val(x) * 32, 261)


def unpack_x32(x):
    return [((x << 5) >> (29 * i)) & MASK for i in range(8)] + [(x << 5) >> 232]


def pack_lt2p(a):
    assert all(x <= MASK for x in a[:8]) and val(a) < 2 * P, "pack29_lt2p needs a normalised value below 2p"
    return val(a) % P


def run_lowered(words, cols, consts, num_cols):
    """cols: canonical R-form integers (one row), consts likewise; returns the canonical accumulator."""
    consts_rp = [(c * 32) % P for c in consts]
    st = []
    acc = unpack(0)
    tmp = {}
    prev_tee = None
    for pc in range(len(words) // 3):
        w0, a, b = (int(x) for x in words[3 * pc:3 * pc + 3])
        op = w0 & 0xff
        has_mem = op == Q_PUSH_COL or K_ADD_COL <= op <= K_FOLD_COL
        if has_mem:
            if a >= num_cols:
                assert prev_tee != a - num_cols, "intermediate read back by the instruction right behind its TEE (prefetch hazard)"
                mem = tmp[a - num_cols]
            else:
                mem = cols[(a, b)]
            assert mem < P
        prev_tee = a if op == Q_TEE_TMP else None
        if w0 & 0x100:
            st[-1] = settle(st[-1])
        if w0 & 0x200:
            st[-2] = settle(st[-2])
        if w0 & 0x400:
            st[-1] = normalize29(st[-1])
        if w0 & 0x800:
            st[-2] = normalize29(st[-2])
        if w0 & 0x1000:
            st[-1] = settle8(st[-1])
        if w0 & 0x2000:
            st[-2] = settle8(st[-2])
        if op == Q_PUSH_COL:
            st.append(unpack(mem))
        elif op == Q_PUSH_CONST:
            st.append(unpack(consts[a]))
        elif op == Q_ADD:
            y = st.pop(); st[-1] = add29(st[-1], y)
        elif op == Q_SUB:
            y = st.pop(); st[-1] = normalize29(sub29k(2, st[-1], y, normalised_after=True))
        elif op == Q_MUL:
            y = st.pop(); st[-1] = mul29(st[-1], shl5(y))
        elif op == Q_NEG:
            st[-1] = normalize29(sub29k(2, unpack(0), st[-1], normalised_after=True))
        elif op == Q_SQUARE:
            st[-1] = mul29(st[-1], shl5(st[-1]))
        elif op == Q_DOUBLE:
            st[-1] = add29(st[-1], st[-1])
        elif op == Q_FOLD:
            acc = add29(mul29(acc, unpack(consts_rp[a])), st.pop())
        elif op == Q_MUL_CONST:
            st[-1] = mul29(st[-1], unpack(consts_rp[a]))
        elif op == Q_ADD_CONST:
            st[-1] = add29(st[-1], unpack(consts[a]))
        elif op == Q_TEE_TMP:
            tmp[a] = pack_lt2p(st[-1])
        elif op == K_ADD_COL:
            st[-1] = add29(st[-1], unpack(mem))
        elif op == K_SUB_COL:
            st[-1] = sub29k(2, st[-1], unpack(mem), normalised_after=False)
        elif op == K_RSUB_COL:
            st[-1] = normalize29(sub29k(2, unpack(mem), st[-1], normalised_after=True))
        elif op == K_MUL_COL:
            st[-1] = mul29(st[-1], unpack_x32(mem))
        elif op == K_FOLD_COL:
            acc = add29(mul29(acc, unpack(consts_rp[w0 >> 16])), unpack(mem))
        elif op == K_NOP:
            pass
        else:
            raise AssertionError(f"unknown lowered opcode {op}")
        assert all(x < (1 << 31) for s_ in st for x in s_), "stack limbs must stay below 2^31"
    assert not st
    acc = normalize29(acc)
    assert val(acc) < 64 * P
    return val(acc) % P


def run_plain(prog, cols, consts):
    """the caller's program over the integers mod p; values are R-form (x R), products divide by R"""
    rinv = pow(R, -1, P)
    st, acc, tmp = [], 0, {}
    for op, a, b in prog:
        if op == Q_PUSH_COL: st.append(cols[(a, b)])
        elif op == Q_PUSH_CONST: st.append(consts[a])
        elif op == Q_ADD: y = st.pop(); st[-1] = (st[-1] + y) % P
        elif op == Q_SUB: y = st.pop(); st[-1] = (st[-1] - y) % P
        elif op == Q_MUL: y = st.pop(); st[-1] = st[-1] * y * rinv % P
        elif op == Q_NEG: st[-1] = -st[-1] % P
        elif op == Q_SQUARE: st[-1] = st[-1] * st[-1] * rinv % P
        elif op == Q_DOUBLE: st[-1] = 2 * st[-1] % P
        elif op == Q_FOLD: acc = (acc * consts[a] * rinv + st.pop()) % P
        elif op == Q_MUL_CONST: st[-1] = st[-1] * consts[a] * rinv % P
        elif op == Q_ADD_CONST: st[-1] = (st[-1] + consts[a]) % P
        elif op == Q_TEE_TMP: tmp[a] = st[-1]
        elif op == Q_PUSH_TMP: st.append(tmp[a])
    return acc


def lower(prog, num_cols, fuse=1):
    lib = binding.lib()
    words = np.array([w for ins in prog for w in ins], dtype=np.uint32)
    n_out, depth = ctypes.c_uint32(), ctypes.c_int()
    cap = 3 * (2 * len(prog) + 8)
    out = np.zeros(cap, dtype=np.uint32)
    rc = lib.zk_host_quotient_lower(words.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(len(prog)), ctypes.c_uint32(num_cols), ctypes.c_int(fuse),
                                    out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap), ctypes.byref(n_out), ctypes.byref(depth))
    assert rc == 0
    return out[:3 * n_out.value], depth.value


# ---- random programs ------------------------------------------------------------------------------
def random_expr(rng, ncols, nconsts, depth, defined_tmp, next_tmp, out):
    """appends postfix code of a random expression; may park and re-read intermediates"""
    r = rng.random()
    if depth == 0 or r < 0.25:
        k = rng.random()
        if k < 0.7:
            out.append((Q_PUSH_COL, rng.randrange(ncols), rng.choice([0, 0, 1, -1 & 0xffffffff, 2])))
        elif k < 0.85 or not defined_tmp:
            out.append((Q_PUSH_CONST, rng.randrange(nconsts), 0))
        else:
            out.append((Q_PUSH_TMP, rng.choice(sorted(defined_tmp)), 0))
        return
    if r < 0.75:
        random_expr(rng, ncols, nconsts, depth - 1, defined_tmp, next_tmp, out)
        random_expr(rng, ncols, nconsts, depth - 1, defined_tmp, next_tmp, out)
        out.append((rng.choice([Q_ADD, Q_SUB, Q_MUL, Q_MUL, Q_ADD]), 0, 0))
    elif r < 0.87:
        random_expr(rng, ncols, nconsts, depth - 1, defined_tmp, next_tmp, out)
        op = rng.choice([Q_NEG, Q_SQUARE, Q_DOUBLE, Q_MUL_CONST, Q_ADD_CONST])
        out.append((op, rng.randrange(nconsts) if op in (Q_MUL_CONST, Q_ADD_CONST) else 0, 0))
    else:
        random_expr(rng, ncols, nconsts, depth - 1, defined_tmp, next_tmp, out)
        slot = next_tmp[0] if (not defined_tmp or rng.random() < 0.7) else rng.choice(sorted(defined_tmp))      # slots may be re-used
        next_tmp[0] = max(next_tmp[0], slot + 1)
        out.append((Q_TEE_TMP, slot, 0))
        defined_tmp.add(slot)


def random_program(rng, ncols, nconsts, statements, depth):
    prog, defined, nxt = [], set(), [0]
    for _ in range(statements):
        random_expr(rng, ncols, nconsts, depth, defined, nxt, prog)
        prog.append((Q_FOLD, rng.randrange(nconsts), 0))
    return prog


def col_values(rng, prog, kind):
    vals = {}
    for op, a, b in prog:
        if op == Q_PUSH_COL and (a, b) not in vals:
            vals[(a, b)] = {"max": P - 1, "zero": 0, "one": R % P}.get(kind, None)
            if vals[(a, b)] is None:
                vals[(a, b)] = rng.choice([P - 1, P - 2, 0, 1, R % P, rng.randrange(P), rng.randrange(P)])
    return vals


@pytest.mark.parametrize("fuse", [1, 0])
def test_lowered_programs_match_plain_evaluation_and_keep_every_bound(fuse):
    rng = random.Random(20260924 + fuse)
    for trial in range(160):
        ncols, nconsts = rng.randrange(1, 7), rng.randrange(1, 4)
        prog = random_program(rng, ncols, nconsts, statements=rng.randrange(1, 6), depth=rng.randrange(1, 6))
        words, depth = lower(prog, ncols, fuse)
        for kind in ("max", "mixed", "mixed", "zero", "one"):
            cols = col_values(rng, prog, kind)
            consts = [rng.choice([P - 1, 1, R % P, rng.randrange(P)]) for _ in range(nconsts)]
            want = run_plain(prog, cols, consts)
            got = run_lowered(words, cols, consts, ncols)
            assert got == want, (trial, kind, prog)


def test_gate_shapes_lower_without_stack_traffic():
    # q * (a * b - c)  and  q * (a + b - c(+1)): the shapes every PLONKish circuit is full of
    prog = [(Q_PUSH_COL, 0, 0), (Q_PUSH_COL, 1, 0), (Q_PUSH_COL, 2, 0), (Q_MUL, 0, 0), (Q_PUSH_COL, 3, 0), (Q_SUB, 0, 0), (Q_MUL, 0, 0), (Q_FOLD, 0, 0),
            (Q_PUSH_COL, 4, 0), (Q_PUSH_COL, 1, 0), (Q_PUSH_COL, 2, 0), (Q_ADD, 0, 0), (Q_PUSH_COL, 3, 1), (Q_SUB, 0, 0), (Q_MUL, 0, 0), (Q_FOLD, 0, 0)]
    words, depth = lower(prog, 5)
    ops = [int(w) for w in words[0::3]]
    assert depth == 1
    assert [o & 0xff for o in ops] == [Q_PUSH_COL, K_MUL_COL, K_SUB_COL, K_MUL_COL, Q_FOLD, Q_PUSH_COL, K_ADD_COL, K_SUB_COL, K_MUL_COL, Q_FOLD]
    assert all(o & 0x300 == 0 for o in ops), "no value of these gates needs settling"
    # a linear combination  sum_j v^j p_j  is one instruction per polynomial
    prog = [(Q_PUSH_COL, j, 0) if i % 2 == 0 else (Q_FOLD, 0, 0) for j in range(5) for i in range(2)]
    words, depth = lower(prog, 5)
    assert [int(w) & 0xff for w in words[0::3]] == [K_FOLD_COL] * 5 and depth == 0


def test_parked_intermediate_is_not_prefetched_behind_its_tee():
    # x TEE 0; PUSH_TMP 0; MUL  (a square through the parking slot): the read must not sit right behind the write
    prog = [(Q_PUSH_COL, 0, 0), (Q_PUSH_COL, 1, 0), (Q_ADD, 0, 0), (Q_TEE_TMP, 0, 0), (Q_PUSH_TMP, 0, 0), (Q_MUL, 0, 0), (Q_FOLD, 0, 0)]
    for fuse in (0, 1):
        words, _ = lower(prog, 2, fuse)
        ops = [int(w) & 0xff for w in words[0::3]]
        t = ops.index(Q_TEE_TMP)
        assert ops[t + 1] == K_NOP
        cols = {(0, 0): P - 1, (1, 0): P - 1}
        assert run_lowered(words, cols, [R % P], 2) == run_plain(prog, cols, [R % P])


def test_malformed_programs_are_refused_not_executed():
    lib = binding.lib()
    for prog in ([(Q_ADD, 0, 0)], [(Q_PUSH_COL, 0, 0), (Q_MUL, 0, 0)], [(Q_FOLD, 0, 0)], [(99, 0, 0)], [(K_MUL_COL, 0, 0)]):
        words = np.array([w for ins in prog for w in ins], dtype=np.uint32)
        n_out, depth = ctypes.c_uint32(), ctypes.c_int()
        rc = lib.zk_host_quotient_lower(words.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(len(prog)), ctypes.c_uint32(4), ctypes.c_int(1),
                                        None, ctypes.c_size_t(0), ctypes.byref(n_out), ctypes.byref(depth))
        assert rc != 0, prog


# ---- the class programs of round 5: one weighted sum (csrc/prover.hip assemble_grouped) -----------------------------------
def test_grouped_class_programs_keep_every_bound():
    """a class program as the prover assembles it since round 5 -- terms under a shared single-column factor collected, weights
    y^(K-1-i) as constants, one closing FOLD -- lowered and executed limb by limb with every column at p - 1, 0, 1 and random:
    long chains of added products are what the settle bits of lower_bounds have to keep inside the limits"""
    import test_quotient_classes as tqc
    rng = random.Random(77)
    lib_mod = type("Z", (), {"lib": staticmethod(binding.lib)})
    S, G = 3, 40
    col = lambda i, rot=0: (Q_PUSH_COL, i, rot)
    terms, cons = [], []
    for g in range(G):                                           # q_s (a b), q_s c, q'_s (a + b - c(+1)): the gate classes of the headline circuit
        q, a, b_, c = col(g % S), col(S + 3 * g), col(S + 3 * g + 1), col(S + 3 * g + 2)
        terms.append([q, a, b_, (Q_MUL, 0, 0), (Q_MUL, 0, 0)]); cons.append(3 * g)
        terms.append([q, c, (Q_NEG, 0, 0), (Q_MUL, 0, 0)]); cons.append(3 * g)
        terms.append([col((g + 1) % S), a, b_, (Q_ADD, 0, 0), (Q_PUSH_COL, S + 3 * g + 2, 1), (Q_SUB, 0, 0), (Q_MUL, 0, 0)]); cons.append(3 * g + 1)
    terms.append([col(S + 1), (Q_NEG, 0, 0)]); cons.append(3 * G + 1)         # a remainder column, weight one
    K = 3 * G + 2
    ncols = S + 3 * G
    prog = tqc.group_terms(lib_mod, terms, cons, K)
    assert prog is not None
    # constant references (0xFFFC0000 + g = y^g, 0xFFFF0004 = one) -> indices, as the prover's concretise does
    y = rng.randrange(P)
    refs = sorted({a for op, a, b in prog if op in (Q_MUL_CONST, Q_ADD_CONST, Q_FOLD, Q_PUSH_CONST)})
    index = {r: i for i, r in enumerate(refs)}
    rinv = pow(R, -1, P)
    def const_of(r):
        if r >= 0xFFFC0000 and r < 0xFFFD0000:
            v = R % P                                            # y^g in R form: (y R)^g / R^(g-1)
            for _ in range(r - 0xFFFC0000):
                v = v * y * rinv % P
            return v
        assert r == 0xFFFF0004
        return R % P
    consts = [const_of(r) for r in refs]
    conc = [(op, index[a] if op in (Q_MUL_CONST, Q_ADD_CONST, Q_FOLD, Q_PUSH_CONST) else a, b) for op, a, b in prog]
    words, depth = lower(conc, ncols)
    assert depth <= 16
    for kind in ("max", "mixed", "mixed", "zero", "one"):
        cols = col_values(rng, conc, kind)
        assert run_lowered(words, cols, consts, ncols) == run_plain(conc, cols, consts), kind
