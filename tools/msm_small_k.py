"""Commit batches at small k (the Keccak configuration is k = 18): per-kernel-group times for witness-like columns (small values
with a few random blinding rows at the end) and dense columns, both MSM paths.  usage: python tools/msm_small_k.py [k]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkevm_circuits_amd as z
from zkevm_circuits_amd import plonk
import bench_proof as bp

k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << k
ctx = z.Context(0)
srs = ctx.srs_setup_with_s(k, np.frombuffer(plonk.fr_mont_bytes(0xC0FFEE), dtype=np.uint64).copy())
rng = np.random.default_rng(3)
def dense_col():
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1); return a
def small_col(bits, blind):
    m = bp.to_mont_gpu(ctx, bp.small_to_limbs(rng.integers(0, 1 << bits, size=n, dtype=np.uint64)))     # Montgomery images, converted on the device
    if blind:
        m[n - blind:] = dense_col()[:blind]
    return m
sets = {"dense": [dense_col() for _ in range(8)], "bits+58 blinding rows": [small_col(1, 58) for _ in range(8)], "bits, no blinding": [small_col(1, 0) for _ in range(8)],
        "bytes+58 blinding rows": [small_col(8, 58) for _ in range(8)]}
for name, cols in sets.items():
    bufs = [ctx.to_device(c) for c in cols]
    ptrs = [b.ptr for b in bufs]
    for hint in (0, 1):
        ctx.commit_batch(srs, ptrs, n, lagrange=True, narrow=[hint] * 8)
        ctx.prof_reset(); ctx.prof_enable(True)
        t0 = time.perf_counter()
        ctx.commit_batch(srs, ptrs, n, lagrange=True, narrow=[hint] * 8)
        dt = (time.perf_counter() - t0) / 8
        ctx.prof_enable(False)
        prof = {nm: ctx.prof_get(nm) for nm in ctx.prof_names()}
        print(f"k={k} {name} hint={hint}: {dt * 1e3:.3f} ms per MSM; " + ", ".join(f"{nm} {ms / max(c, 1):.3f}" for nm, (ms, c) in sorted(prof.items())))
    for b in bufs: b.free()
ctx.close()
