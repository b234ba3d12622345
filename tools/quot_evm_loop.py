"""The evaluator on the program it exists for: the compiled top-class program of the EVM-style constraint system
(bench_proof.evm_block, EVM_DEFAULT: 5 258 constraints of degree <= 9 over 160 step columns; csrc/class_compile.hpp) over random
columns at 2^k rows, `reps` launches -- the profiling loop of round 6 (tools/quot_loop.py is round 5's, two gate shapes).
usage: python tools/quot_evm_loop.py [k] [reps] [distinct column buffers, 0 = one per column]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_proof as bp  # noqa: E402
from zkevm_circuits_amd import binding, plonk  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nbuf = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = 1 << k
p = dict(bp.EVM_DEFAULT)
c = plonk.Circuit(10, num_fixed=1, num_advice=bp.evm_step_columns(p), num_instance=0, blinding_factors=5)
bp.evm_block(c, 0, c.fixed_col(0), p)
lib = binding.lib()
blob = c.cs_blob()
E = c.extended_k() - c.k
os.environ.setdefault("ZK_QUOTIENT_SPLIT", "0")          # one class: the whole constraint system in one program
summ = np.zeros(8 + 8 * (E + 1), dtype=np.uint32)
cnt = ctypes.c_uint32()
ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
assert lib.zk_host_quotient_plan(blob, ctypes.c_size_t(len(blob)), ptr(summ), ctypes.c_size_t(summ.size), ctypes.c_uint32(E), None, ctypes.c_size_t(0), ctypes.byref(cnt)) == 0
words = np.zeros(3 * cnt.value, dtype=np.uint32)
assert lib.zk_host_quotient_plan(blob, ctypes.c_size_t(len(blob)), ptr(summ), ctypes.c_size_t(summ.size), ctypes.c_uint32(E), ptr(words), ctypes.c_size_t(words.size), ctypes.byref(cnt)) == 0
prog = words.reshape(-1, 3).copy()
col_ix, const_ix = {}, {}
for ins in prog:
    op, a = int(ins[0]), int(ins[1])
    if op == 1:
        ins[1] = col_ix.setdefault(a, len(col_ix))
    elif op in (2, 9, 10, 11):
        ins[1] = const_ix.setdefault(a, len(const_ix))
products = int(np.isin(prog[:, 0], (5, 7, 9, 10)).sum())
ctx = binding.Context(0)
rng = np.random.default_rng(1)
nb = nbuf if nbuf else len(col_ix)
bufs = []
for _ in range(nb):
    v = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    v[:, 3] &= np.uint64((1 << 60) - 1)
    bufs.append(ctx.to_device(v))
ptrs = [bufs[i % nb].ptr for i in range(len(col_ix))]
consts = rng.integers(0, 1 << 62, size=(max(len(const_ix), 1), 4), dtype=np.uint64)
consts[:, 3] &= np.uint64((1 << 60) - 1)
out = ctx.alloc(n * 32)
ctx.quotient_eval(prog, ptrs, consts, k, k, out)
ctx.sync()
ctx.timer_start()
for _ in range(reps):
    ctx.quotient_eval(prog, ptrs, consts, k, k, out)
ms = ctx.timer_stop_ms() / reps
loads = int((prog[:, 0] == 1).sum())
print(f"k={k} evm class program: {len(prog)} instructions, {products} products, {loads} column reads ({len(col_ix)} distinct (column, rotation) operands over {nb} buffers) per row: "
      f"{ms:.2f} ms per launch = {products * n / ms / 1e6:.1f} G products/s, {loads * n * 32 / ms / 1e9:.2f} TB/s of operand loads")
ctx.close()
