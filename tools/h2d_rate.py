"""Raw host-to-device rate from page-locked memory: back-to-back 32 MiB copies on one stream, alone and with a 192-byte copy behind
each (the blinding rows of a witness column travel that way).  usage: python tools/h2d_rate.py"""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so.7")
def ck(e):
    assert e == 0, e
N = 32 << 20
cols = 64
h = ctypes.c_void_p(); ck(hip.hipHostMalloc(ctypes.byref(h), ctypes.c_size_t(N * 4), 0))
d = ctypes.c_void_p(); ck(hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(N * cols)))
s = ctypes.c_void_p(); ck(hip.hipStreamCreateWithFlags(ctypes.byref(s), 1))
def run(tail, chunks=1):
    ck(hip.hipStreamSynchronize(s))
    t0 = time.perf_counter()
    for c in range(cols):
        src = h.value + (c % 4) * N
        dst = d.value + c * N
        step = (N - tail) // chunks
        for j in range(chunks):
            ck(hip.hipMemcpyAsync(ctypes.c_void_p(dst + j * step), ctypes.c_void_p(src + j * step), ctypes.c_size_t(step), 1, s))
        if tail:
            ck(hip.hipMemcpyAsync(ctypes.c_void_p(dst + N - tail), ctypes.c_void_p(src + N - tail), ctypes.c_size_t(tail), 1, s))
    ck(hip.hipStreamSynchronize(s))
    dt = time.perf_counter() - t0
    return cols * N / dt / 1e9, dt / cols * 1e3
for name, tail, chunks in (("32 MiB copies", 0, 1), ("32 MiB + 192 B tail", 192, 1), ("2 x 16 MiB", 0, 2), ("32 MiB copies", 0, 1)):
    run(tail, chunks)
    gbs, ms = run(tail, chunks)
    print(f"{name:24s}: {gbs:6.1f} GB/s, {ms:.3f} ms per column")

# a witness column that lives in ordinary (pageable) memory: upload as it is, or page-lock it first (hipHostRegister) and upload
import numpy as np
cols_np = [np.random.default_rng(i).integers(0, 1 << 62, size=N // 8, dtype=np.uint64) for i in range(8)]
ck(hip.hipStreamSynchronize(s))
t0 = time.perf_counter()
for c, a in enumerate(cols_np):
    ck(hip.hipMemcpyAsync(ctypes.c_void_p(d.value + c * N), ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(N), 1, s))
ck(hip.hipStreamSynchronize(s))
t_page = (time.perf_counter() - t0) / len(cols_np)
t0 = time.perf_counter()
for a in cols_np:
    ck(hip.hipHostRegister(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(N), 0))
t_reg = (time.perf_counter() - t0) / len(cols_np)
t0 = time.perf_counter()
for c, a in enumerate(cols_np):
    ck(hip.hipMemcpyAsync(ctypes.c_void_p(d.value + c * N), ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(N), 1, s))
ck(hip.hipStreamSynchronize(s))
t_pin = (time.perf_counter() - t0) / len(cols_np)
t0 = time.perf_counter()
for a in cols_np:
    ck(hip.hipHostUnregister(ctypes.c_void_p(a.ctypes.data)))
t_unreg = (time.perf_counter() - t0) / len(cols_np)
print(f"pageable column: {t_page * 1e3:.3f} ms upload; register {t_reg * 1e3:.3f} ms + upload {t_pin * 1e3:.3f} ms + unregister {t_unreg * 1e3:.3f} ms")
