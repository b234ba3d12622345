"""Dense commit batches at small k through the graph-replay path (no profiling events: they switch it off): ms per MSM for
batches of 8 and 16 columns.  usage: [ZK_MSM_GRAPH_PIPES=N] python tools/msm_graph_pipes.py [k]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkevm_circuits_amd as z
from zkevm_circuits_amd import plonk

k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << k
ctx = z.Context(0)
srs = ctx.srs_setup_with_s(k, np.frombuffer(plonk.fr_mont_bytes(0xC0FFEE), dtype=np.uint64).copy())
rng = np.random.default_rng(3)
def dense_col():
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1); return a
bufs = [ctx.to_device(dense_col()) for _ in range(16)]
for cnt in (8, 16):
    ptrs = [b.ptr for b in bufs[:cnt]]
    for basis in (True, False):
        ctx.commit_batch(srs, ptrs, n, lagrange=basis)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.commit_batch(srs, ptrs, n, lagrange=basis)
            best = min(best, (time.perf_counter() - t0) / cnt)
        print(f"k={k} pipes={os.environ.get('ZK_MSM_GRAPH_PIPES', '4')} batch of {cnt} dense columns, {'lagrange' if basis else 'coefficient'} basis: {best * 1e3:.3f} ms per MSM")
ctx.close()
