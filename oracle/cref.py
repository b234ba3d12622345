"""ctypes loader for oracle/liboracle.so (TEST INFRASTRUCTURE ONLY -- see oracle/c/oracle.c).

All arrays are numpy ``uint64`` arrays in the halo2curves in-memory layout:
  field elements  (n, 4)   Montgomery limbs, little-endian
  G1Affine        (n, 8)   x || y
  G1 (Jacobian)   (n, 12)  x || y || z
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from . import bn254

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
FR, FQ = 0, 1


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "c", "oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def usable_cpus() -> int:
    """CPUs this process may really use: scheduler affinity capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:   # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def lib():
    global _LIB
    if _LIB is None:
        # libgomp reads these when it is loaded.  On a box whose cgroup quota is far below the
        # visible core count, one spinning thread per visible core turns every parallel region
        # into seconds of barrier time: size the team to what can run, and sleep at barriers.
        user = os.environ.get("OMP_NUM_THREADS")
        team = int(user) if user and user.isdigit() else usable_cpus()
        os.environ.setdefault("OMP_NUM_THREADS", str(team))
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        _LIB = ctypes.CDLL(build())
        _LIB.orc_set_num_threads(team)   # the OpenMP runtime may already be loaded (torch) with its own team size
    return _LIB


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


# ---------------------------------------------------------------- int <-> limb conversion
def ints_to_limbs(vals) -> np.ndarray:
    """Plain integers -> (n,4) u64 limbs (no Montgomery conversion)."""
    out = np.empty((len(vals), 4), dtype=np.uint64)
    buf = b"".join(int(v).to_bytes(32, "little") for v in vals)
    out[:] = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)
    return out


def limbs_to_ints(a: np.ndarray):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    raw = a.tobytes()
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(a.shape[0])]


def to_mont(vals, which=FR) -> np.ndarray:
    a = ints_to_limbs(vals)
    o = np.empty_like(a)
    lib().orc_fe_to_mont_vec(which, _p(a), _p(o), ctypes.c_size_t(a.shape[0]))
    return o


def from_mont(a: np.ndarray, which=FR):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    o = np.empty_like(a)
    lib().orc_fe_from_mont_vec(which, _p(a), _p(o), ctypes.c_size_t(a.shape[0]))
    return limbs_to_ints(o)


def affine_to_mont(points) -> np.ndarray:
    """List of affine int tuples / None -> (n,8) Montgomery array (identity = zeros)."""
    flat = []
    for p in points:
        flat += [0, 0] if p is None else [p[0], p[1]]
    return to_mont(flat, FQ).reshape(-1, 8)


def affine_from_mont(a: np.ndarray):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 8)
    vals = from_mont(a.reshape(-1, 4), FQ)
    out = []
    for i in range(a.shape[0]):
        x, y = vals[2 * i], vals[2 * i + 1]
        out.append(None if (x == 0 and y == 0) else (x, y))
    return out


# ---------------------------------------------------------------- field vectors
def fe_binop(name: str, which: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    o = np.empty_like(a)
    getattr(lib(), f"orc_fe_{name}_vec")(which, _p(a), _p(b), _p(o), ctypes.c_size_t(a.shape[0]))
    return o


def fe_inv(which: int, a: np.ndarray) -> np.ndarray:
    o = np.empty_like(a)
    lib().orc_fe_inv_vec(which, _p(a), _p(o), ctypes.c_size_t(a.shape[0]))
    return o


def rand_fr_stream(seed: int, n: int) -> np.ndarray:
    o = np.empty((n, 4), dtype=np.uint64)
    lib().orc_rand_fr_stream(ctypes.c_uint64(seed), _p(o), ctypes.c_size_t(n))
    return o


# ---------------------------------------------------------------- NTT & polynomial helpers
def fr_const(v: int) -> np.ndarray:
    return to_mont([v % bn254.R_MOD], FR)


def best_fft(a: np.ndarray, omega: int, log_n: int) -> np.ndarray:
    """In-place on a copy; a is (n,4) Montgomery; omega is a plain integer."""
    a = np.ascontiguousarray(a.copy())
    w = fr_const(omega)
    lib().orc_best_fft(_p(a), _p(w), ctypes.c_uint(log_n))
    return a


def scale(a: np.ndarray, s: int) -> np.ndarray:
    a = np.ascontiguousarray(a.copy())
    lib().orc_scale_vec(_p(a), _p(fr_const(s)), ctypes.c_size_t(a.shape[0]))
    return a


def distribute_powers(a: np.ndarray, g: int) -> np.ndarray:
    a = np.ascontiguousarray(a.copy())
    lib().orc_distribute_powers(_p(a), _p(fr_const(g)), ctypes.c_size_t(a.shape[0]))
    return a


def ifft(a: np.ndarray, log_n: int) -> np.ndarray:
    om_inv = bn254.fr_inv(bn254.omega_for_k(log_n))
    return scale(best_fft(a, om_inv, log_n), bn254.fr_inv(1 << log_n))


def eval_polynomial(c: np.ndarray, x: int) -> int:
    o = np.empty((1, 4), dtype=np.uint64)
    lib().orc_eval_polynomial(_p(c), ctypes.c_size_t(c.shape[0]), _p(fr_const(x)), _p(o))
    return from_mont(o)[0]


def kate_division(c: np.ndarray, z: int) -> np.ndarray:
    q = np.empty((c.shape[0] - 1, 4), dtype=np.uint64)
    lib().orc_kate_division(_p(c), ctypes.c_size_t(c.shape[0]), _p(fr_const(z)), _p(q))
    return q


def batch_invert(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a.copy())
    lib().orc_batch_invert(_p(a), ctypes.c_size_t(a.shape[0]))
    return a


def prefix_product(a: np.ndarray) -> np.ndarray:
    z = np.empty_like(a)
    lib().orc_prefix_product(_p(a), _p(z), ctypes.c_size_t(a.shape[0]))
    return z


def prefix_sum(a: np.ndarray) -> np.ndarray:
    z = np.empty_like(a)
    lib().orc_prefix_sum(_p(a), _p(z), ctypes.c_size_t(a.shape[0]))
    return z


# ---------------------------------------------------------------- G1
def g1_jac_add(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    o = np.empty_like(a)
    lib().orc_g1_jac_add_vec(_p(a), _p(b), _p(o), ctypes.c_size_t(a.shape[0]))
    return o


def g1_jac_madd(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    o = np.empty_like(a)
    lib().orc_g1_jac_madd_vec(_p(a), _p(b), _p(o), ctypes.c_size_t(a.shape[0]))
    return o


def g1_jac_double(a: np.ndarray) -> np.ndarray:
    o = np.empty_like(a)
    lib().orc_g1_jac_double_vec(_p(a), _p(o), ctypes.c_size_t(a.shape[0]))
    return o


def g1_to_affine(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 12)
    o = np.empty((a.shape[0], 8), dtype=np.uint64)
    lib().orc_g1_to_affine_vec(_p(a), _p(o), ctypes.c_size_t(a.shape[0]))
    return o


def g1_mul(points: np.ndarray, scalars: np.ndarray) -> np.ndarray:
    o = np.empty_like(points)
    lib().orc_g1_mul_vec(_p(points), _p(scalars), _p(o), ctypes.c_size_t(points.shape[0]))
    return o


def srs_powers(s: int, n: int) -> np.ndarray:
    g = np.empty((n, 8), dtype=np.uint64)
    lib().orc_srs_powers(_p(fr_const(s)), _p(g), ctypes.c_size_t(n))
    return g


def hash_to_curve_points(seed: int, n: int) -> np.ndarray:
    """n pseudo-random affine G1 points (unknown discrete logs): SURVEY 8(d) config 2's second base set"""
    g = np.empty((n, 8), dtype=np.uint64)
    lib().orc_hash_to_curve_points(ctypes.c_uint64(seed), _p(g), ctypes.c_size_t(n))
    return g


def best_multiexp(scalars: np.ndarray, bases: np.ndarray, threads: int = 0) -> np.ndarray:
    """halo2 best_multiexp restated; returns (8,) affine Montgomery."""
    if threads <= 0:
        threads = lib().orc_num_threads()
    o = np.empty((1, 8), dtype=np.uint64)
    lib().orc_best_multiexp(_p(scalars), _p(bases), ctypes.c_size_t(scalars.shape[0]), ctypes.c_int(threads), _p(o))
    return o[0]


def num_threads() -> int:
    return lib().orc_num_threads()
