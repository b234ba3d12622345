"""zk_fr_batch_invert: microseconds per call at several sizes (events around 20 calls).  usage: python tools/bi_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkevm_circuits_amd as z
ctx = z.Context(0)
rng = np.random.default_rng(1)
for k in (10, 14, 16, 18, 20, 22):
    n = 1 << k
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
    d = ctx.to_device(a)
    ctx.fr_batch_invert(d, n); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20): ctx.fr_batch_invert(d, n)
    ctx.sync()
    print(f"k={k}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per call")
    d.free()
ctx.close()
