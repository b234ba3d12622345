import sys, time
sys.path.insert(0, '.')
import numpy as np
import importlib
z = importlib.import_module('zkevm_circuits_amd.binding')
from oracle import cref
ctx = z.Context(0)
for k in (8, 12, 17, 20, 22):
    n = 1 << k
    a = cref.rand_fr_stream(5, n)
    d = ctx.to_device(a)
    for rep in range(3):
        t0 = time.perf_counter(); ctx.ntt(d, k); ctx.sync(); t1 = time.perf_counter()
        print(k, rep, "ntt %.3f ms" % ((t1 - t0) * 1e3))
    t0 = time.perf_counter(); r = cref.best_fft(a, 5, k); t1 = time.perf_counter()
    print(k, "oracle %.3f ms" % ((t1 - t0) * 1e3))
