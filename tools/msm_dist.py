"""MSM cost per column by value distribution ("witness-like" = 60 % zero / 30 % < 2^16 / 10 % uniform per cell), on the
paths a caller can ask for: wall time per MSM inside a pipelined batch, kernel groups, a lone commitment.
usage: python tools/msm_dist.py [k] [columns] [dist,dist,...] [hint,hint]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkevm_circuits_amd as z
from zkevm_circuits_amd import plonk
import bench_proof as bp

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = 1 << k
ctx = z.Context(0)
srs = ctx.srs_setup_with_s(k, np.frombuffer(plonk.fr_mont_bytes(0xC0FFEE), dtype=np.uint64).copy())
rng = np.random.default_rng(3)


def uniform_limbs(m):
    a = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(m, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a


def column(dist):
    if dist == "uniform":
        return uniform_limbs(n)                     # Montgomery images of uniform values are uniform
    small = rng.integers(0, 1 << 16, size=n, dtype=np.uint64)
    if dist == "small16":
        return bp.to_mont_gpu(ctx, bp.small_to_limbs(small))
    frac_large = {"survey_60_30_10": 0.10, "third_uniform": 1 / 3, "one_percent": 0.01, "tenth_percent": 0.001}[dist]
    u = rng.random(n)
    small[u < 0.6] = 0
    col = bp.to_mont_gpu(ctx, bp.small_to_limbs(small))
    big = np.flatnonzero(u >= 1 - frac_large)
    col[big] = uniform_limbs(big.size)
    return col


dists = sys.argv[3].split(",") if len(sys.argv) > 3 else ("uniform", "third_uniform", "survey_60_30_10", "one_percent", "tenth_percent", "small16")
hints = [int(h) for h in sys.argv[4].split(",")] if len(sys.argv) > 4 else (0, 1)
for dist in dists:
    bufs = [ctx.to_device(column(dist)) for _ in range(ncol)]
    ptrs = [b.ptr for b in bufs]
    for hint in hints:
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
        lone = []
        for i in range(4):
            t1 = time.perf_counter()
            ctx.commit_batch(srs, ptrs[i:i + 1], n, lagrange=True, narrow=[hint])
            lone.append(time.perf_counter() - t1)
        print(f"k={k} {dist:16s} hint={hint}: {dt * 1e3:.3f} ms per MSM in a batch of {ncol}, lone {sorted(lone)[1] * 1e3:.3f} ms; "
              + ", ".join(f"{nm} {ms / max(c, 1):.3f}" for nm, (ms, c) in sorted(prof.items())), flush=True)
    for b in bufs:
        b.free()
ctx.close()
