"""Expression-evaluator loop for profiling: G groups of the two gate shapes of the SuperCircuit-shape bench
(q (a b - c), q (a + b - c(+1))) over 3 G + S columns at 2^k rows, `reps` launches.
usage: python tools/quot_loop.py [k] [groups] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkevm_circuits_amd import binding

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = 1 << k
ctx = binding.Context(0)
rng = np.random.default_rng(1)
ncols = 3 * G + 8
cols = []
for c in range(ncols):
    v = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    v[:, 3] &= np.uint64((1 << 60) - 1)
    cols.append(ctx.to_device(v))
prog = []
for g in range(G):
    a, b_, c = 3 * g, 3 * g + 1, 3 * g + 2
    q0, q1 = 3 * G + (g % 4) * 2, 3 * G + (g % 4) * 2 + 1
    prog += [(1, q0, 0), (1, a, 0), (1, b_, 0), (5, 0, 0), (1, c, 0), (4, 0, 0), (5, 0, 0), (9, 0, 0)]
    prog += [(1, q1, 0), (1, a, 0), (1, b_, 0), (3, 0, 0), (1, c, 1), (4, 0, 0), (5, 0, 0), (9, 0, 0)]
prog = np.array(prog, dtype=np.uint32)
consts = np.array([[5, 6, 7, 1]], dtype=np.uint64)
out = ctx.alloc(n * 32)
ptrs = [c.ptr for c in cols]
ctx.quotient_eval(prog, ptrs, consts, k, k, out)
ctx.sync()
ctx.timer_start()
for _ in range(reps):
    ctx.quotient_eval(prog, ptrs, consts, k, k, out)
ms = ctx.timer_stop_ms() / reps
reads = 3 * G + G + 8          # a, b, c, c(+1) per group + selectors
print(f"k={k} groups={G} fuse={os.environ.get('ZK_QUOTIENT_FUSE', '1')}: {ms:.3f} ms per launch, {3 * G} products + {2 * G} folds per row, "
      f"{(reads + 1) * n * 32 / ms / 1e6:.0f} GB/s algorithmic")
ctx.close()
