"""Profiling workload: `reps` forward NTTs of size 2^k on one resident buffer (rocprofv3 target)."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zkevm_circuits_amd as z

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = z.Context(0)
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 60, size=(1 << k, 4), dtype=np.uint64)
d = ctx.to_device(a)
for _ in range(reps):
    ctx.ntt(d, k)
ctx.sync()
