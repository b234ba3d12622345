"""Per-transform time of batched 2^k transforms as the prover issues them: forward / inverse NTTs and coset transforms over `cols`
columns, several columns per launch, on a rotating set of buffers larger than the Infinity Cache.
usage: python tools/ntt_batch_time.py [k] [cols] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkevm_circuits_amd as z
from zkevm_circuits_amd import plonk

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n = 1 << k
ctx = z.Context(0)
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 60, size=(n, 4), dtype=np.uint64)
src = [ctx.to_device(a) for _ in range(cols)]
dst = [ctx.alloc(n * 32) for _ in range(cols)]
g = np.frombuffer(plonk.fr_mont_bytes(7), dtype=np.uint64).copy()


def timed(fn, label):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / (reps * cols)
    print(f"k={k} cols={cols} {label}: {dt * 1e6:.1f} us per transform (ZK_NTT_LIMB={os.environ.get('ZK_NTT_LIMB', '0')})", flush=True)


timed(lambda: ctx.ntt_batch(src, k), "forward ntt_batch")
timed(lambda: ctx.ntt_batch(src, k, inverse=True), "inverse ntt_batch")
timed(lambda: ctx.coeff_to_coset_batch(src, k, g, dst), "coeff_to_coset_batch")
for i in range(0, cols, 4):      # four columns per call, as the prover's stages do
    pass
timed(lambda: [ctx.coeff_to_coset_batch(src[i:i + 4], k, g, dst[i:i + 4]) for i in range(0, cols, 4)], "coeff_to_coset_batch x4")
