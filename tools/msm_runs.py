"""MSM 2^20 over columns made of runs of equal field-sized values (what a permutation running product looks like on rows
without copies): per-kernel-group times for the three commit hints.  usage: python tools/msm_runs.py [run_length]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkevm_circuits_amd as z
from zkevm_circuits_amd import plonk

k = 20; n = 1 << k
run = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ctx = z.Context(0)
srs = ctx.srs_setup_with_s(k, np.frombuffer(plonk.fr_mont_bytes(0xC0FFEE), dtype=np.uint64).copy())
rng = np.random.default_rng(3)
def runs_column(seed):
    r = np.random.default_rng(seed)
    nv = n // run + 2
    vals = r.integers(0, 1 << 62, size=(nv, 4), dtype=np.uint64); vals[:, 3] &= np.uint64((1 << 60) - 1)
    cuts = np.sort(r.integers(0, n, size=nv - 1)); idx = np.searchsorted(cuts, np.arange(n), side="right")
    return vals[idx]
cols = [ctx.to_device(runs_column(i)) for i in range(8)]
dense = [ctx.to_device(rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64) & np.uint64((1 << 60) - 1)) for _ in range(8)]
for name, bufs in (("runs", cols), ("dense", dense)):
    for hint in (0, 1, 2):
        ptrs = [b.ptr for b in bufs]
        ctx.commit_batch(srs, ptrs, n, lagrange=True, narrow=[hint] * 8)
        ctx.prof_reset(); ctx.prof_enable(True)
        t0 = time.perf_counter()
        ctx.commit_batch(srs, ptrs, n, lagrange=True, narrow=[hint] * 8)
        dt = (time.perf_counter() - t0) / 8
        ctx.prof_enable(False)
        prof = {nm: ctx.prof_get(nm) for nm in ctx.prof_names()}
        print(f"{name} run={run} hint={hint}: {dt * 1e3:.3f} ms per MSM; " + ", ".join(f"{nm} {ms / max(c, 1):.3f}" for nm, (ms, c) in sorted(prof.items())))
ctx.close()
