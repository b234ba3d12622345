"""Which library call leaves a HIP error behind (hipPeekAtLastError after every step)?"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkevm_circuits_amd as z
from oracle import cref
hip = ctypes.CDLL("libamdhip64.so.7")
hip.hipGetErrorString.restype = ctypes.c_char_p
def peek(tag):
    e = hip.hipPeekAtLastError()
    print(f"{tag:40s} hipPeekAtLastError = {e} {hip.hipGetErrorString(e).decode() if e else ''}")
ctx = z.Context(0); peek("ctx")
k, n = 12, 1 << 12
srs = ctx.srs_setup_with_s(k, cref.fr_const(5)); peek("srs_setup")
col = ctx.to_device(cref.rand_fr_stream(3, n)); peek("to_device")
ctx.prof_reset(); ctx.prof_enable(2); peek("prof_enable 2")
ctx.commit(srs, col, n, lagrange=True); peek("commit")
ctx.ntt(col, k); peek("ntt")
ctx.prof_enable(False); names = ctx.prof_names(); peek("prof_names")
ctx.prof_reset(); ctx.prof_enable(True)
ctx.commit(srs, col, n, lagrange=True); peek("commit (prof all)")
out = ctx.alloc(n * 32)
prog = np.array([(1, 0, 0), (1, 0, 1), (5, 0, 0), (9, 0, 0)], dtype=np.uint32)
ctx.quotient_eval(prog, [col.ptr], cref.fr_const(1).reshape(1, 4), k, k, out); peek("quotient_eval")
ctx.prof_enable(False); print(ctx.prof_names()); peek("prof_names 2")
print(ctx.prof_get_bytes("quotient_eval")); peek("prof_get_bytes")
srs.destroy(); peek("srs.destroy")
uid = ctx.comm_unique_id(); peek("unique id")
ctx.comm_init(uid, 0, 1); peek("comm_init")
ctx.comm_destroy(); ctx.close()
