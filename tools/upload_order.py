"""Does the advice-phase upload rate depend on what the context did before?  usage: python tools/upload_order.py [prelude]
prelude = none | msm (a pipelined commit batch first: creates the side streams before the copy stream)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkevm_circuits_amd as z
import bench_proof as bp
from zkevm_circuits_amd import plonk

prelude = sys.argv[1] if len(sys.argv) > 1 else "none"
if "torch" in prelude:
    import torch
    torch.cuda.set_device(0)
    _t = torch.zeros(64, dtype=torch.uint8, device="cuda")
ctx = z.Context(0)
if "msm" in prelude:
    k = 20
    s_mont = np.frombuffer(plonk.fr_mont_bytes(0xC0FFEE), dtype=np.uint64).copy()
    srs = ctx.srs_setup_with_s(k, s_mont)
    rng = np.random.default_rng(1)
    a = rng.integers(0, 1 << 62, size=(1 << k, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
    cols = [ctx.to_device(a) for _ in range(4)]
    if "prof" in prelude:
        ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(3):
        ctx.commit_batch(srs, [c.ptr for c in cols], 1 << k, lagrange=True)
        if "ntt" in prelude:
            for c in cols:
                ctx.ntt(c, k, inverse=True)
    if "prof" in prelude:
        ctx.prof_enable(False); _ = {nm: ctx.prof_get(nm) for nm in ctx.prof_names()}
    if "lone" in prelude:
        for i in range(6):
            ctx.commit(srs, cols[i % 4], 1 << k, lagrange=True)
    if "dl" in prelude:
        _ = srs.download_g_lagrange()
    for c in cols:
        c.free()
    srs.destroy()
circ, blob, adv_m, inst_m, inst = bp.build_shape(ctx, 20, 1000, 150, 150, 100, 9)
rec = bp.proof_bench(ctx, circ.k, circ, blob, adv_m, inst_m, inst, shplonk=True, repeat=2, verify=False, pinned=True)
print(prelude, os.environ.get("ZK_EAGER_COPY_STREAM", "0"), rec["create_proof_s"])
ctx.close()
