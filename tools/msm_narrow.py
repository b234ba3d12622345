"""Small-valued columns on the per-window MSM path (what the advice phase of a wide circuit commits): wall time per MSM inside
a pipelined batch, and the kernel groups.  usage: python tools/msm_narrow.py [k] [bits] [columns]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkevm_circuits_amd as z
from zkevm_circuits_amd import plonk
import bench_proof as bp

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ncol = int(sys.argv[3]) if len(sys.argv) > 3 else 32
n = 1 << k
ctx = z.Context(0)
srs = ctx.srs_setup_with_s(k, np.frombuffer(plonk.fr_mont_bytes(0xC0FFEE), dtype=np.uint64).copy())
rng = np.random.default_rng(3)
cols = [bp.to_mont_gpu(ctx, bp.small_to_limbs(rng.integers(0, 1 << bits, size=n, dtype=np.uint64))) for _ in range(ncol)]
bufs = [ctx.to_device(c) for c in cols]
ptrs = [b.ptr for b in bufs]
for hint in (1, 0):
    ctx.commit_batch(srs, ptrs, n, lagrange=True, narrow=[hint] * ncol)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.commit_batch(srs, ptrs, n, lagrange=True, narrow=[hint] * ncol)
    dt = (time.perf_counter() - t0) / (3 * ncol)
    ctx.prof_reset(); ctx.prof_enable(True)
    ctx.commit_batch(srs, ptrs, n, lagrange=True, narrow=[hint] * ncol)
    ctx.prof_enable(False)
    prof = {nm: ctx.prof_get(nm) for nm in ctx.prof_names()}
    print(f"k={k} values < 2^{bits} hint={hint}: {dt * 1e3:.3f} ms per MSM in a batch of {ncol}; " + ", ".join(f"{nm} {ms / max(c, 1):.3f}" for nm, (ms, c) in sorted(prof.items())))
ctx.close()
