"""Host-side circuit description for the prover: a small mirror of halo2's ``ConstraintSystem`` /
``Expression`` / permutation ``Assembly`` that serialises to the flat "pk blob" consumed by
``zk_pk_create`` (witness / constraint-system export format).

In production the Rust shim fills the same blob from ``halo2_proofs::plonk::ConstraintSystem``
(gates -> postfix programs, ``permutation::keygen::Assembly`` -> sigma columns, fixed columns in
Lagrange form); this module lets tests and benches build circuits without Rust.

Expressions are tiny ASTs with operator overloading:  ``a * b - c``, ``q * (x.rot(1) - x - 1)``.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Sequence, Tuple

import numpy as np

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
FR_GENERATOR = 7
FR_S = 28
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (R_MOD - 1) >> FR_S, R_MOD)
FR_DELTA = pow(FR_GENERATOR, 1 << FR_S, R_MOD)

FIXED, ADVICE, INSTANCE = 0, 1, 2
Q_PUSH_COL, Q_PUSH_CONST, Q_ADD, Q_SUB, Q_MUL, Q_NEG = 1, 2, 3, 4, 5, 6
Q_TEE_TMP, Q_PUSH_TMP = 12, 13          # intermediates shared between gates (csrc/quotient.hip)
BLOB_MAGIC, BLOB_VERSION = 0x4B505A4B, 3
C_CHAL0 = 0xFFFD0000   # abstract constant reference of user challenge i (csrc/prover.hip)


def fr_mont_bytes(v: int) -> bytes:
    return (((v % R_MOD) << 256) % R_MOD).to_bytes(32, "little")


def column_to_mont(values: Sequence[int]) -> np.ndarray:
    """Plain integers -> (n, 4) u64 Montgomery limbs (the layout every ABI call uses)."""
    buf = b"".join(fr_mont_bytes(v) for v in values)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


# ------------------------------------------------------------------------------------ expressions
class Expr:
    def __add__(self, o): return Bin(Q_ADD, self, wrap(o))
    def __radd__(self, o): return Bin(Q_ADD, wrap(o), self)
    def __sub__(self, o): return Bin(Q_SUB, self, wrap(o))
    def __rsub__(self, o): return Bin(Q_SUB, wrap(o), self)
    def __mul__(self, o): return Bin(Q_MUL, self, wrap(o))
    def __rmul__(self, o): return Bin(Q_MUL, wrap(o), self)
    def __neg__(self): return Neg(self)


class Col(Expr):
    def __init__(self, ctype: int, index: int, rotation: int = 0):
        self.ctype, self.index, self.rotation = ctype, index, rotation

    def rot(self, r: int) -> "Col":
        return Col(self.ctype, self.index, self.rotation + r)

    def degree(self): return 1


class Const(Expr):
    def __init__(self, value: int):
        self.value = value % R_MOD

    def degree(self): return 0


class Challenge(Expr):
    """halo2 ``Challenge``: a transcript challenge usable in gates of later phases."""

    def __init__(self, index: int):
        self.index = index

    def degree(self): return 0


class Bin(Expr):
    def __init__(self, op, a, b):
        self.op, self.a, self.b = op, a, b

    def degree(self):
        return self.a.degree() + self.b.degree() if self.op == Q_MUL else max(self.a.degree(), self.b.degree())


class Neg(Expr):
    def __init__(self, a):
        self.a = a

    def degree(self): return self.a.degree()


def wrap(x) -> Expr:
    return x if isinstance(x, Expr) else Const(int(x))


def colref(ctype: int, index: int) -> int:
    return (ctype << 24) | index


def expr_identifier(e: Expr) -> str:
    """halo2 `Expression::identifier()` [EXT-RECALL plonk/circuit.rs]: the string the mv-lookup fork keys its `lookups_map`
    (a BTreeMap, so arguments come out in the order of these strings) with -- tables that differ get different keys,
    and for the column-only tables of the fixtures the order is the order of `fixed[i][rot]` strings."""
    if isinstance(e, Col):
        return f"{ {FIXED: 'fixed', ADVICE: 'advice', INSTANCE: 'instance'}[e.ctype]}[{e.index}][{e.rotation}]"
    if isinstance(e, Const):
        return f"0x{e.value:064x}"
    if isinstance(e, Challenge):
        return f"challenge[{e.index}]"
    if isinstance(e, Neg):
        return "(-" + expr_identifier(e.a) + ")"
    return "(" + expr_identifier(e.a) + {Q_ADD: "+", Q_SUB: "-", Q_MUL: "*"}[e.op] + expr_identifier(e.b) + ")"


def expr_tokens(e: Expr) -> List[str]:
    """postfix tokens of an expression, leaves in the order halo2's closures would query them (depth first, left to right):
    a:<col>:<rot> f:<col>:<rot> i:<col>:<rot> k:<hex constant> c:<challenge> + - * n(egate)"""
    if isinstance(e, Col):
        return [f"{ {FIXED: 'f', ADVICE: 'a', INSTANCE: 'i'}[e.ctype]}:{e.index}:{e.rotation}"]
    if isinstance(e, Const):
        return [f"k:{e.value:x}"]
    if isinstance(e, Challenge):
        return [f"c:{e.index}"]
    if isinstance(e, Neg):
        return expr_tokens(e.a) + ["n"]
    return expr_tokens(e.a) + expr_tokens(e.b) + [{Q_ADD: "+", Q_SUB: "-", Q_MUL: "*"}[e.op]]


def expr_from_tokens(tokens: Sequence[str]) -> Expr:
    st: List[Expr] = []
    for t in tokens:
        if t in "+-*":
            b_, a_ = st.pop(), st.pop()
            st.append(Bin({"+": Q_ADD, "-": Q_SUB, "*": Q_MUL}[t], a_, b_))
        elif t == "n":
            st.append(Neg(st.pop()))
        else:
            kind, *rest = t.split(":")
            if kind in "afi":
                st.append(Col({"f": FIXED, "a": ADVICE, "i": INSTANCE}[kind], int(rest[0]), int(rest[1])))
            elif kind == "k":
                st.append(Const(int(rest[0], 16)))
            elif kind == "c":
                st.append(Challenge(int(rest[0])))
            else:
                raise ValueError(f"bad token {t!r}")
    assert len(st) == 1
    return st[0]


class Lookup:
    """halo2 `mv_lookup::Argument`: one table tuple and one or more input tuples looked up in it
    (`inputs_expressions: Vec<Vec<Expression>>`; `chunk_lookups()` merges the inputs that share a
    table and splits them again so that the argument's degree stays within the circuit's)."""

    def __init__(self, name: str, table: Sequence[Expr], inputs: Sequence[Sequence[Expr]]):
        self.name, self.table, self.inputs = name, list(table), [list(i) for i in inputs]
        assert all(len(i) == len(self.table) for i in self.inputs)

    def required_degree(self) -> int:
        """halo2 mv_lookup::Argument::required_degree: the grand-sum identity
        l_active * (tau * prod phi_i * (phi(wX) - phi(X)) - ...) has degree table + sum(inputs) + 2."""
        table_degree = max(e.degree() for e in self.table)
        inputs_degree = sum(max(e.degree() for e in i) for i in self.inputs)
        return max(3 + len(self.inputs), table_degree + inputs_degree + 2)


class Circuit:
    """Shape + fixed assignment + copy constraints of one PLONKish circuit (mirror of halo2's
    `ConstraintSystem` + keygen `Assembly`).  Column queries are registered in call order, as
    halo2's `query_advice / query_fixed / enable_equality` do: that order is the order of the
    evaluations in the proof, so it is part of the exported key blob."""

    def __init__(self, k: int, num_fixed: int, num_advice: int, num_instance: int, blinding_factors: int = 5, shape_only: bool = False):
        """shape_only: no fixed assignment is kept (a verifier needs the constraint system alone; at k = 25 the cell lists of a
        production aggregation circuit would not fit) -- such a circuit cannot be turned into a key blob."""
        self.k, self.n = k, 1 << k
        self.F, self.A, self.I = num_fixed, num_advice, num_instance
        self.bf = blinding_factors
        self.u = self.n - self.bf - 1          # rows [0, u) are usable
        self.gates: List[Expr] = []
        self.lookups: List[Lookup] = []
        self.lookups_map: Dict[str, Lookup] = {}      # halo2 `lookups_map`: table identifier -> tracker, until chunk_lookups()
        self.minimum_degree = 1
        self.perm_cols: List[Tuple[int, int]] = []
        self.copies: List[Tuple[Tuple[int, int, int], Tuple[int, int, int]]] = []
        self.fixed = None if shape_only else [[0] * self.n for _ in range(num_fixed)]
        self.consts: List[int] = []
        self._const_index: Dict[int, int] = {}
        self.advice_phase = [0] * num_advice      # halo2 FirstPhase = 0, SecondPhase = 1, ...
        self.challenge_phase: List[int] = []      # challenge i is squeezed after this phase
        self.advice_queries: List[Tuple[int, int]] = []      # (column, rotation), registration order
        self.fixed_queries: List[Tuple[int, int]] = []
        self.instance_queries: List[Tuple[int, int]] = []
        # configure-time calls in call order (what `Circuit::configure` of a Rust circuit must replay to arrive at the same
        # ConstraintSystem: query registration order = evaluation order, enable_equality order = permutation column order)
        self.ops: List[tuple] = []

    # -- columns
    def fixed_col(self, i, rot=0): return Col(FIXED, i, rot)
    def advice_col(self, i, rot=0): return Col(ADVICE, i, rot)
    def instance_col(self, i, rot=0): return Col(INSTANCE, i, rot)

    def _register(self, e: Expr):
        """halo2 query_*_index: first use of a (column, rotation) pair appends it to the query list"""
        if isinstance(e, Col):
            lst = {FIXED: self.fixed_queries, ADVICE: self.advice_queries, INSTANCE: self.instance_queries}[e.ctype]
            if (e.index, e.rotation) not in lst:
                lst.append((e.index, e.rotation))
        elif isinstance(e, Neg):
            self._register(e.a)
        elif isinstance(e, Bin):
            self._register(e.a)
            self._register(e.b)

    # -- constraints
    def add_gate(self, e: Expr):
        self._register(e)
        self.gates.append(e)
        self.ops.append(("gate", e))

    def add_lookup(self, inputs: Sequence[Expr], tables: Sequence[Expr], name: str = "lookup"):
        """one lookup argument with one input tuple (what `lookup_any` yields when nothing is merged)"""
        assert len(inputs) == len(tables)
        for e in list(inputs) + list(tables):
            self._register(e)
        self.lookups.append(Lookup(name, tables, [inputs]))
        self.ops.append(("lookup", name, list(inputs), list(tables)))

    def lookup_any(self, name: str, inputs: Sequence[Expr], tables: Sequence[Expr]):
        """halo2 `ConstraintSystem::lookup_any` of the mv-lookup fork: lookups into the same table
        expressions are collected under one tracker; `chunk_lookups()` turns the trackers into
        arguments [REF zkevm-circuits/src/evm_circuit/execution.rs:978-1014]."""
        assert len(inputs) == len(tables)
        for e in list(inputs) + list(tables):
            self._register(e)
        self.ops.append(("lookup", name, list(inputs), list(tables)))
        ident = "".join(expr_identifier(t) for t in tables)      # upstream: table_expressions_identifier, the BTreeMap key
        if ident in self.lookups_map:
            self.lookups_map[ident].inputs.append(list(inputs))
        else:
            self.lookups_map[ident] = Lookup(name, tables, [inputs])

    def chunk_lookups(self):
        """halo2 `ConstraintSystem::chunk_lookups` [REF zkevm-circuits/src/super_circuit/test.rs:59],
        [REF aggregator/src/aggregation/config.rs:223]: fix the circuit degree at (next power of two of
        the largest gate / single-lookup degree - 1) + 1, then greedily pack the inputs of every table into
        as few arguments as that degree allows."""
        if not self.lookups_map:
            return self
        max_gate_degree = max([g.degree() for g in self.gates] + [0])
        max_single = 0
        for v in self.lookups_map.values():
            base = max(3, max(e.degree() for e in v.table) + 2)
            max_single = max(max_single, base + max(max(e.degree() for e in i) for i in v.inputs))
        required = max(max_gate_degree, max_single)
        required = 1 << max(required - 2, 0).bit_length()            # (required - 1).next_power_of_two()
        self.minimum_degree = max(self.minimum_degree, required + 1)
        for key in sorted(self.lookups_map):
            v = self.lookups_map[key]
            args = [Lookup(v.name, v.table, [])]
            for inp in v.inputs:
                cur = max(e.degree() for e in inp)
                for a in args:
                    if a.required_degree() + cur <= self.minimum_degree:
                        a.inputs.append(list(inp))
                        break
                else:
                    args.append(Lookup(v.name, v.table, [inp]))
            self.lookups += [a for a in args if a.inputs]
        self.lookups_map = {}
        return self

    def challenge_usable_after(self, phase: int) -> "Challenge":
        self.challenge_phase.append(phase)
        return Challenge(len(self.challenge_phase) - 1)

    def num_phases(self) -> int:
        return max([0] + self.advice_phase + self.challenge_phase) + 1

    def enable_equality(self, ctype: int, index: int):
        if (ctype, index) not in self.perm_cols:
            self.perm_cols.append((ctype, index))
            self._register(Col(ctype, index, 0))       # halo2: enable_equality queries the column at Rotation::cur()
            self.ops.append(("enable_equality", ctype, index))

    def copy(self, a: Tuple[int, int, int], b: Tuple[int, int, int]):
        """(ctype, index, row) == (ctype, index, row)"""
        for c in (a, b):
            self.enable_equality(c[0], c[1])
            assert c[2] < self.u, "copy constraint on an unusable row"
        self.copies.append((a, b))

    # -- derived shape
    def degree(self) -> int:
        """halo2 `ConstraintSystem::degree`: permutation argument 3, every lookup's required
        degree, every gate polynomial, and the minimum degree set by chunk_lookups()."""
        assert not self.lookups_map, "call chunk_lookups() before the circuit is used"
        d = 3
        for lk in self.lookups:
            d = max(d, lk.required_degree())
        for g in self.gates:
            d = max(d, g.degree())
        return max(d, self.minimum_degree)

    def extended_k(self) -> int:
        ek = self.k
        while (1 << ek) < self.n * (self.degree() - 1):
            ek += 1
        return ek

    def omega(self) -> int:
        return pow(FR_ROOT_OF_UNITY, 1 << (FR_S - self.k), R_MOD)

    # -- compilation
    def _const(self, v: int) -> int:
        if v not in self._const_index:
            self._const_index[v] = len(self.consts)
            self.consts.append(v)
        return self._const_index[v]

    def compile(self, e: Expr) -> List[Tuple[int, int, int]]:
        out: List[Tuple[int, int, int]] = []

        def go(x: Expr):
            if isinstance(x, Col):
                out.append((Q_PUSH_COL, colref(x.ctype, x.index), x.rotation & 0xFFFFFFFF))
            elif isinstance(x, Const):
                out.append((Q_PUSH_CONST, self._const(x.value), 0))
            elif isinstance(x, Challenge):
                out.append((Q_PUSH_CONST, C_CHAL0 + x.index, 0))
            elif isinstance(x, Neg):
                go(x.a)
                out.append((Q_NEG, 0, 0))
            else:
                go(x.a)
                go(x.b)
                out.append((x.op, 0, 0))
        go(e)
        return out

    def compile_gates_cse(self) -> List[List[Tuple[int, int, int]]]:
        """Every gate as a postfix program, with the common-subexpression elimination halo2's
        ``GraphEvaluator`` performs over the whole gate list: a sub-expression that holds a product
        and occurs more than once (in the same or in different gates) is computed at its first
        occurrence, parked with TEE_TMP, and read back with PUSH_TMP afterwards.  The gates are
        evaluated in order in one launch, so an intermediate written by gate i is visible to gate
        j > i.  Same values as ``compile`` gate by gate -- only cheaper."""
        def key(x):
            if isinstance(x, Col):
                return ("c", x.ctype, x.index, x.rotation)
            if isinstance(x, Const):
                return ("k", x.value)
            if isinstance(x, Challenge):
                return ("h", x.index)
            if isinstance(x, Neg):
                return ("n", key(x.a))
            return ("b", x.op, key(x.a), key(x.b))

        def has_mul(x):
            if isinstance(x, Bin):
                return x.op == Q_MUL or has_mul(x.a) or has_mul(x.b)
            return isinstance(x, Neg) and has_mul(x.a)

        uses: Dict[tuple, int] = {}

        def count(x):
            kx = key(x)
            uses[kx] = uses.get(kx, 0) + 1
            if uses[kx] > 1:
                return                      # its children are evaluated once only
            if isinstance(x, Neg):
                count(x.a)
            elif isinstance(x, Bin):
                count(x.a)
                count(x.b)
        for g in self.gates:
            count(g)
        slot: Dict[tuple, int] = {}
        progs = []
        for g in self.gates:
            out: List[Tuple[int, int, int]] = []

            def go(x):
                kx = key(x)
                if kx in slot:
                    out.append((Q_PUSH_TMP, slot[kx], 0))
                    return
                if isinstance(x, Col):
                    out.append((Q_PUSH_COL, colref(x.ctype, x.index), x.rotation & 0xFFFFFFFF))
                elif isinstance(x, Const):
                    out.append((Q_PUSH_CONST, self._const(x.value), 0))
                elif isinstance(x, Challenge):
                    out.append((Q_PUSH_CONST, C_CHAL0 + x.index, 0))
                elif isinstance(x, Neg):
                    go(x.a)
                    out.append((Q_NEG, 0, 0))
                else:
                    go(x.a)
                    go(x.b)
                    out.append((x.op, 0, 0))
                if uses.get(kx, 0) > 1 and has_mul(x):
                    slot[kx] = len(slot)
                    out.append((Q_TEE_TMP, slot[kx], 0))
            go(g)
            progs.append(out)
        return progs

    def halo2_blinding_factors(self) -> int:
        """halo2 `ConstraintSystem::blinding_factors` [EXT-RECALL plonk/circuit.rs]: max(3, most distinct rotations any advice
        column is queried at) + 1 (multi-open) + 1 (h evaluation).  Upstream DERIVES this number; a circuit meant for upstream's
        verifier must be built with it (the fixtures of the GPU suite are free to use more blinding rows)."""
        per_col = [0] * max(self.A, 1)
        for col, _rot in self.advice_queries:
            per_col[col] += 1
        return max(3, max(per_col)) + 2

    # -- T1 kit: the circuit as a text file a stand-alone Rust program replays against UPSTREAM halo2 (shim/t1_standalone)
    def kit_desc(self) -> str:
        """Everything `Circuit::configure` + keygen's `synthesize` of an equivalent Rust circuit need, one item per line:
        shape, phases, the configure-time calls in call order, non-zero fixed cells, copy constraints in call order
        (advice / fixed pairs first, pairs with an instance cell last: a Rust region can only express them in that order)."""
        tname = {FIXED: "fixed", ADVICE: "advice", INSTANCE: "instance"}
        assert not self.lookups_map, "call chunk_lookups() first"
        assert self.bf == self.halo2_blinding_factors(), f"upstream derives blinding_factors = {self.halo2_blinding_factors()} for this circuit, it was built with {self.bf}"
        region = [c for c in self.copies if INSTANCE not in (c[0][0], c[1][0])]
        inst = [c for c in self.copies if INSTANCE in (c[0][0], c[1][0])]
        assert self.copies == region + inst, "copy constraints with instance cells must come last (constrain_instance runs after the region)"
        out = ["zkmi355-t1-kit 1", f"k {self.k}", f"blinding_factors {self.bf}", f"degree {self.degree()}",
               f"fixed {self.F}", "advice " + " ".join(str(p) for p in self.advice_phase) if self.A else "advice",
               f"instance {self.I}", "challenges " + " ".join(str(p) for p in self.challenge_phase)]
        for op in self.ops:
            if op[0] == "gate":
                out.append("gate " + " ".join(expr_tokens(op[1])))
            elif op[0] == "lookup":
                _, name, inputs, tables = op
                out.append(f"lookup {name.replace(' ', '_')} {len(inputs)} " + " | ".join(" ".join(expr_tokens(e)) for e in list(inputs) + list(tables)))
            else:
                out.append(f"enable_equality {tname[op[1]]} {op[2]}")
        for col in range(self.F):
            for row, v in enumerate(self.fixed[col]):
                if v % R_MOD:
                    out.append(f"fixed_cell {col} {row} {v % R_MOD:x}")
        for a, b in self.copies:
            out.append(f"copy {tname[a[0]]} {a[1]} {a[2]} {tname[b[0]]} {b[1]} {b[2]}")
        out.append("end")
        return "\n".join(out) + "\n"

    @classmethod
    def from_kit_desc(cls, text: str) -> "Circuit":
        """Rebuilds the circuit from the file alone (what the Rust side does, restated: the kit's self-check proves that
        the file carries everything)."""
        tnum = {"fixed": FIXED, "advice": ADVICE, "instance": INSTANCE}
        lines = [ln.split() for ln in text.splitlines() if ln.strip()]
        assert lines[0] == ["zkmi355-t1-kit", "1"] and lines[-1] == ["end"]
        head = {ln[0]: ln[1:] for ln in lines[1:8]}
        circ = cls(int(head["k"][0]), int(head["fixed"][0]), len(head["advice"]), int(head["instance"][0]), int(head["blinding_factors"][0]))
        circ.advice_phase = [int(p) for p in head["advice"]]
        circ.challenge_phase = [int(p) for p in head["challenges"]]
        any_lookup = False
        for ln in lines[8:-1]:
            if ln[0] == "gate":
                circ.add_gate(expr_from_tokens(ln[1:]))
            elif ln[0] == "lookup":
                n_in = int(ln[2])
                groups, cur = [], []
                for tok in ln[3:]:
                    if tok == "|":
                        groups.append(cur)
                        cur = []
                    else:
                        cur.append(tok)
                groups.append(cur)
                exprs = [expr_from_tokens(g) for g in groups]
                circ.lookup_any(ln[1], exprs[:n_in], exprs[n_in:])
                any_lookup = True
            elif ln[0] == "enable_equality":
                circ.enable_equality(tnum[ln[1]], int(ln[2]))
            elif ln[0] == "fixed_cell":
                circ.fixed[int(ln[1])][int(ln[2])] = int(ln[3], 16)
            elif ln[0] == "copy":
                circ.copy((tnum[ln[1]], int(ln[2]), int(ln[3])), (tnum[ln[4]], int(ln[5]), int(ln[6])))
            else:
                raise ValueError(f"bad line {ln}")
        if any_lookup:
            circ.chunk_lookups()           # upstream keygen does this itself after configure [EXT-RECALL plonk/keygen.rs create_domain]
        assert circ.degree() == int(head["degree"][0]) and circ.bf == int(head["blinding_factors"][0])
        return circ

    def permutation_mapping(self) -> List[List[tuple]]:
        """halo2 ``permutation::keygen::Assembly::mapping``: mapping[j][i] = (j', i'), the next cell in the cycle of equal
        cells that cell i of permutation column j belongs to (itself when the cell takes part in no copy constraint)."""
        P, n = len(self.perm_cols), self.n
        pos = {c: j for j, c in enumerate(self.perm_cols)}
        mapping = [[(j, i) for i in range(n)] for j in range(P)]
        aux = [[(j, i) for i in range(n)] for j in range(P)]
        sizes = [[1] * n for _ in range(P)]
        for a, b in self.copies:
            lc, lr, rc, rr = pos[(a[0], a[1])], a[2], pos[(b[0], b[1])], b[2]
            if aux[lc][lr] == aux[rc][rr]:
                continue
            lcyc, rcyc = aux[lc][lr], aux[rc][rr]
            if sizes[lcyc[0]][lcyc[1]] < sizes[rcyc[0]][rcyc[1]]:
                lc, lr, rc, rr = rc, rr, lc, lr
                lcyc, rcyc = rcyc, lcyc
            sizes[lcyc[0]][lcyc[1]] += sizes[rcyc[0]][rcyc[1]]
            i = (rc, rr)
            while True:
                aux[i[0]][i[1]] = lcyc
                i = mapping[i[0]][i[1]]
                if i == (rc, rr):
                    break
            mapping[lc][lr], mapping[rc][rr] = mapping[rc][rr], mapping[lc][lr]
        return mapping

    def sigma_columns(self) -> List[List[int]]:
        """halo2 ``permutation::keygen::Assembly``: cycles of equal cells -> sigma_j(omega^i) =
        delta^j' * omega^i' of the next cell in the cycle."""
        P, n = len(self.perm_cols), self.n
        mapping = self.permutation_mapping()
        w = self.omega()
        wp = [1] * n
        for i in range(1, n):
            wp[i] = wp[i - 1] * w % R_MOD
        dp = [pow(FR_DELTA, j, R_MOD) for j in range(P)]
        return [[dp[mapping[j][i][0]] * wp[mapping[j][i][1]] % R_MOD for i in range(n)] for j in range(P)]

    def cs_blob(self, cse: bool = False) -> bytes:
        """The constraint-system part of the key blob (everything but the column data): what the
        default `vk.transcript_repr` of zk_pk_create hashes together with the key's commitments."""
        gates = self.compile_gates_cse() if cse else [self.compile(g) for g in self.gates]
        lookups = [([self.compile(e) for e in lk.table], [[self.compile(e) for e in i] for i in lk.inputs]) for lk in self.lookups]

        def prog(p):
            return struct.pack("<I", len(p)) + b"".join(struct.pack("<III", *ins) for ins in p)

        def queries(q):
            return struct.pack("<I", len(q)) + b"".join(struct.pack("<Ii", c, r) for c, r in q)

        out = [struct.pack("<12I", BLOB_MAGIC, BLOB_VERSION, self.k, self.bf, self.degree(), self.F, self.A, self.I,
                           len(self.perm_cols), len(self.lookups), len(gates), len(self.consts))]
        out.append(struct.pack("<I", len(self.challenge_phase)))
        out += [struct.pack("<I", p) for p in self.advice_phase]
        out += [struct.pack("<I", p) for p in self.challenge_phase]
        out += [queries(self.advice_queries), queries(self.fixed_queries), queries(self.instance_queries)]
        out += [struct.pack("<II", t, i) for t, i in self.perm_cols]
        out += [fr_mont_bytes(c) for c in self.consts]
        out += [prog(g) for g in gates]
        for tabs, inputs in lookups:
            out.append(struct.pack("<II", len(tabs), len(inputs)))
            out += [prog(p) for p in tabs]
            for ins in inputs:
                out += [prog(p) for p in ins]
        return b"".join(out)

    def blob(self, cse: bool = False) -> bytes:
        """Serialise for zk_pk_create (layout documented in INTEGRATION.md).  cse: share
        sub-expressions between gates through the evaluator's intermediates (same proof bytes)."""
        assert self.fixed is not None, "a shape-only circuit has no fixed assignment to export"
        sig = self.sigma_columns()
        out = [self.cs_blob(cse)]
        out += [column_to_mont(col).tobytes() for col in self.fixed]
        out += [column_to_mont(col).tobytes() for col in sig]
        return b"".join(out)
