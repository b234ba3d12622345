"""CPU: where zk_quotient_eval cuts a program into slices (zk_host_quotient_slices; csrc/quotient.hip: plan_slices) and that the cut is sound:
the partial sums of the slices, put together with the products of the slices' fold constants, are the value of the whole program (big-int).
The device side of the same thing is tests/test_gpu_quotient.py::test_sliced_program_matches_oracle."""
import ctypes
import random

import numpy as np
import pytest

from oracle import bn254 as b
from zkevm_circuits_amd import binding

R = b.R_MOD
PUSH_COL, PUSH_CONST, ADD, SUB, MUL, FOLD, MUL_CONST, TEE, PUSH_TMP = 1, 2, 3, 4, 5, 9, 10, 12, 13


def cuts_of(prog, ext_k):
    lib = binding.lib()
    words = np.ascontiguousarray(np.array(prog, dtype=np.uint32).reshape(-1))
    out = np.zeros(len(prog) + 2, dtype=np.uint32)
    cnt = ctypes.c_uint32()
    rc = lib.zk_host_quotient_slices(words.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(len(prog)), ctypes.c_uint32(ext_k), out.ctypes.data_as(ctypes.c_void_p),
                                     ctypes.c_size_t(out.size), ctypes.byref(cnt))
    assert rc == 0
    return [int(x) for x in out[:cnt.value]]


def run(prog, row, consts, acc=0):
    st, tmp = [], {}
    for op, a, _ in prog:
        if op == PUSH_COL: st.append(row[a])
        elif op == PUSH_CONST: st.append(consts[a])
        elif op == ADD: y = st.pop(); st[-1] = (st[-1] + y) % R
        elif op == SUB: y = st.pop(); st[-1] = (st[-1] - y) % R
        elif op == MUL: y = st.pop(); st[-1] = st[-1] * y % R
        elif op == MUL_CONST: st[-1] = st[-1] * consts[a] % R
        elif op == FOLD: acc = (acc * consts[a] + st.pop()) % R
        elif op == TEE: tmp[a] = st[-1]
        elif op == PUSH_TMP: st.append(tmp[a])
    assert not st
    return acc


def terms(rng, n, ncols, park_every=0):
    prog = []
    for t in range(n):
        prog += [(PUSH_COL, rng.randrange(ncols), 0), (PUSH_COL, rng.randrange(ncols), 0), (MUL, 0, 0)]
        if park_every and t % park_every == 1: prog += [(TEE, 0, 0)]
        if park_every and t % park_every == 3: prog += [(PUSH_TMP, 0, 0), (ADD, 0, 0)]
        prog += [(PUSH_COL, rng.randrange(ncols), 0), (SUB, 0, 0), (FOLD, rng.randrange(4), 0)]
    return prog


def test_small_or_read_once_programs_stay_in_one_piece(monkeypatch):
    monkeypatch.delenv("ZK_QUOTIENT_SLICES", raising=False)
    rng = random.Random(1)
    assert cuts_of(terms(rng, 100, 8), 20) == []                # short
    assert cuts_of(terms(rng, 1000, 8), 12) == []               # few rows: their operands are in the caches anyway
    wide = []
    for t in range(1000):                                        # 3 000 operands read once each
        wide += [(PUSH_COL, 3 * t, 0), (PUSH_COL, 3 * t + 1, 0), (MUL, 0, 0), (PUSH_COL, 3 * t + 2, 0), (SUB, 0, 0), (FOLD, 0, 0)]
    assert cuts_of(wide, 20) == []
    monkeypatch.setenv("ZK_QUOTIENT_SLICES", "0")
    assert cuts_of(terms(rng, 2000, 64), 20) == []


def test_cuts_fall_on_top_level_folds_and_never_inside_a_parked_value(monkeypatch):
    monkeypatch.delenv("ZK_QUOTIENT_SLICES", raising=False)
    rng = random.Random(2)
    prog = terms(rng, 3000, 300, park_every=5)                   # 300 columns x 32 B x 327 680 rows in flight = 3 GB: many slices wanted
    cuts = cuts_of(prog, 20)
    assert len(cuts) >= 10 and cuts[0] == 0 and cuts[-1] == len(prog) and cuts == sorted(set(cuts))
    sizes = [y - x for x, y in zip(cuts, cuts[1:])]
    assert max(sizes) <= 2 * (len(prog) / len(sizes)) + 16       # balanced
    alive = False
    bad = set()
    for pc, (op, a, _) in enumerate(prog):                       # positions strictly inside (TEE, last PUSH_TMP]
        if op == TEE: alive = True
        elif alive: bad.add(pc)
        if op == PUSH_TMP: alive = False
    for c in cuts[1:-1]:
        assert prog[c - 1][0] == FOLD and c not in bad
    # the identity the device relies on: acc_out(slice) = acc_in * prod(fold constants of the slice) + P_slice
    consts = [rng.randrange(R) for _ in range(4)]
    row = [rng.randrange(R) for _ in range(300)]
    acc = 0
    for x, y in zip(cuts, cuts[1:]):
        sl = prog[x:y]
        kprod = 1
        for op, a, _ in sl:
            if op == FOLD: kprod = kprod * consts[a] % R
        acc = (acc * kprod + run(sl, row, consts)) % R
    assert acc == run(prog, row, consts)


def test_a_value_parked_across_the_whole_program_forbids_every_cut(monkeypatch):
    monkeypatch.setenv("ZK_QUOTIENT_SLICES", "8")
    rng = random.Random(3)
    prog = [(PUSH_COL, 0, 0), (PUSH_COL, 1, 0), (MUL, 0, 0), (TEE, 0, 0), (FOLD, 0, 0)] + terms(rng, 500, 8) + [(PUSH_TMP, 0, 0), (FOLD, 1, 0)]
    assert cuts_of(prog, 20) == []
    assert len(cuts_of(terms(rng, 500, 8), 20)) == 9            # the same terms without it: the eight slices asked for
