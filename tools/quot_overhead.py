"""Cost of one evaluator instruction by kind: chains of 1000 ADD_COL / MUL_COL / FOLD_COL / MUL_CONST / ADD_CONST over 2^20 rows (ns per lowered
instruction, wave and SIMD).  The column chains re-read 60 columns and are bound by memory (32 B per row and instruction); the MUL_CONST chain has no
memory operand: its time over the bare product (366 ns) is the interpreter's own overhead per instruction (~56 ns).  usage: python tools/quot_overhead.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkevm_circuits_amd import binding
k = 20; n = 1 << k
ctx = binding.Context(0)
rng = np.random.default_rng(1)
ncols = 64
cols = []
for c in range(ncols):
    v = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); v[:, 3] &= np.uint64((1 << 60) - 1)
    cols.append(ctx.to_device(v))
ptrs = [c.ptr for c in cols]
consts = np.array([[5, 6, 7, 1]], dtype=np.uint64)
out = ctx.alloc(n * 32)
def run(prog, label, reps=5):
    prog = np.array(prog, dtype=np.uint32)
    ctx.quotient_eval(prog, ptrs, consts, k, k, out); ctx.sync()
    ctx.timer_start()
    for _ in range(reps): ctx.quotient_eval(prog, ptrs, consts, k, k, out)
    ms = ctx.timer_stop_ms() / reps
    print(f"{label}: {ms:.3f} ms per launch for {len(prog)} caller instructions -> {ms * 1e6 / len(prog) * 1024 / (n / 64):.1f} ns per instruction, wave and SIMD")
N = 1000
# PUSH c0, then (PUSH c_i, ADD) x N  -> lowered: PUSH + N x ADD_COL (+ settles), FOLD
run([(1, 0, 0)] + [x for i in range(N) for x in ((1, 1 + i % 60, 0), (3, 0, 0))] + [(9, 0, 0)], "ADD_COL chain")
# PUSH c0, then (PUSH c_i, MUL) x N  -> MUL_COL chain
run([(1, 0, 0)] + [x for i in range(N) for x in ((1, 1 + i % 60, 0), (5, 0, 0))] + [(9, 0, 0)], "MUL_COL chain")
# (PUSH c_i, FOLD) x N -> FOLD_COL chain
run([x for i in range(N) for x in ((1, 1 + i % 60, 0), (9, 0, 0))], "FOLD_COL chain")
# MUL_CONST chain
run([(1, 0, 0)] + [(10, 0, 0)] * N + [(9, 0, 0)], "MUL_CONST chain")
# ADD_CONST chain
run([(1, 0, 0)] + [(11, 0, 0)] * N + [(9, 0, 0)], "ADD_CONST chain")
ctx.close()
