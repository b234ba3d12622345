import sys, time
sys.path.insert(0, '.')
import numpy as np, importlib
z = importlib.import_module('zkevm_circuits_amd.binding')
from oracle import cref
ctx = z.Context(0)
for k in (14, 20, 21):
    n = 1 << k
    a = cref.rand_fr_stream(5, n)
    d = ctx.to_device(a)
    ctx.fr_batch_invert(d, n); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10): ctx.fr_batch_invert(d, n)
    ctx.sync(); t1 = time.perf_counter()
    print(k, "batch_invert %.3f ms" % ((t1 - t0) * 100))
