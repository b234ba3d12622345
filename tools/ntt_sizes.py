"""NTT time per transform at several sizes, on a rotating set of buffers larger than the Infinity Cache; single launches and
batched launches (zk_ntt_batch).  usage: python tools/ntt_sizes.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkevm_circuits_amd as z

ctx = z.Context(0)
rng = np.random.default_rng(1)
for k in (16, 18, 20, 22, 23, 24):
    n = 1 << k
    nbuf = max(2, min(64, (1 << 30) // (n * 32)))          # >= 1 GiB working set where it fits
    a = rng.integers(0, 1 << 60, size=(n, 4), dtype=np.uint64)
    bufs = [ctx.to_device(a) for _ in range(nbuf)]
    reps = max(nbuf, 32)
    for b in bufs: ctx.ntt(b, k)
    ctx.sync(); ctx.timer_start()
    for i in range(reps): ctx.ntt(bufs[i % nbuf], k)
    single = ctx.timer_stop_ms() / reps
    ctx.ntt_batch(bufs, k)
    ctx.sync(); ctx.timer_start()
    rounds = max(1, reps // nbuf)
    for _ in range(rounds): ctx.ntt_batch(bufs, k)
    batched = ctx.timer_stop_ms() / (rounds * nbuf)
    print(f"k={k}: {single * 1e3:8.1f} us per transform alone, {batched * 1e3:8.1f} us batched ({nbuf} buffers), "
          f"{1.5 * n * k / (batched * 1e-3) / 1e9:6.1f} Gfield-op/s, {64 * n / (batched * 1e-3) / 1e9:6.1f} GB/s algorithmic")
    for b in bufs: b.free()
ctx.close()
