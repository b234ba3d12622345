"""CPU: the kernels' field and curve arithmetic (csrc/ff29.hip.hpp, csrc/ec29.hip.hpp: 9 x 29-bit limbs, R' = 2^261
Montgomery form, lazily reduced) compiled for the host and checked against big-int arithmetic and the oracle's group law
(oracle/bn254.py) -- values, limb bounds and the documented result bounds, with operands at the edges of every
precondition.  The GPU tests check the kernels end to end; this pins the arithmetic they are made of without a GPU.
"""
import ctypes
import os
import random
import shutil
import subprocess

import pytest

from oracle import bn254 as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, R = o.P_MOD, o.R_MOD
MOD = {0: P, 1: R}
RP = 1 << 261
M29 = (1 << 29) - 1
U9 = ctypes.c_uint32 * 9
U8 = ctypes.c_uint32 * 8
U18 = ctypes.c_uint32 * 18
U36 = ctypes.c_uint32 * 36


@pytest.fixture(scope="module")
def h(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("harness") / "host_arith.so"
    subprocess.run([hipcc, "--cuda-host-only", "-O2", "-std=c++17", "-shared", "-fPIC",
                    os.path.join(ROOT, "tests", "host_arith_harness.hip"), "-o", str(out)], check=True, timeout=300)
    return ctypes.CDLL(str(out))


def limbs(v, top_extra=True):
    """normalised limbs of v (< 2^261 + room in the top limb)"""
    out = [(v >> (29 * i)) & M29 for i in range(8)]
    out.append(v >> 232)
    assert out[8] < 1 << 32
    return out


def val(l):
    return sum(int(x) << (29 * i) for i, x in enumerate(l))


def unnormalise(l, rng, maxlimb):
    """same value, limbs pushed up to < maxlimb by moving carries down (limb i+1 gives 2^29 to limb i)"""
    l = list(l)
    for i in range(7, -1, -1):
        room = (maxlimb - 1 - l[i]) >> 29
        take = min(room, l[i + 1], rng.randrange(0, 4))
        l[i] += take << 29
        l[i + 1] -= take
    return l


def normalised(l):
    return all(int(x) <= M29 for x in l[:8])


def edge_values(mod, rng, hi_mult):
    """values in [0, hi_mult * mod): the edges and random ones"""
    top = hi_mult * mod
    vs = [0, 1, mod - 1, mod, mod + 1, 2 * mod - 1, top - 1, top // 2]
    vs += [rng.randrange(top) for _ in range(24)]
    return [v for v in vs if v < top]


@pytest.mark.parametrize("which", [0, 1])
def test_unpack_pack_round_trip(h, which):
    rng = random.Random(1 + which)
    mod = MOD[which]
    for v in [0, 1, mod - 1, (1 << 256) - 1, 1 << 255] + [rng.randrange(1 << 256) for _ in range(50)]:
        out = U9()
        h.h_unpack29(which, U8(*[(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]), out)
        assert val(out) == v and normalised(out)
    for v in edge_values(mod, rng, 4):           # pack29: canonical representative of a normalised value < 4p
        o8 = U8()
        h.h_pack29(which, U9(*limbs(v)), o8)
        assert sum(int(x) << (32 * i) for i, x in enumerate(o8)) == v % mod
    for v in edge_values(mod, rng, 2):
        o8 = U8()
        h.h_pack29_lt2p(which, U9(*limbs(v)), o8)
        assert sum(int(x) << (32 * i) for i, x in enumerate(o8)) == v % mod


@pytest.mark.parametrize("which", [0, 1])
def test_montgomery_product_and_square(h, which):
    """mul29(a, b) = a b 2^-261 mod p, normalised, below a b / 2^261 + p; first operand limbs up to 2^30, a b < 2^261 p."""
    rng = random.Random(7 + which)
    mod = MOD[which]
    rinv = pow(RP, -1, mod)
    a_vals = edge_values(mod, rng, 16) + [(1 << 258) - 1]
    b_vals = edge_values(mod, rng, 9) + [(1 << 256) - 1]
    for a in a_vals:
        for b in rng.sample(b_vals, 6) + [b_vals[0], b_vals[2], b_vals[-1]]:
            if a * b >= RP * mod:
                continue
            la = unnormalise(limbs(a), rng, 1 << 30)
            out = U9()
            h.h_mul29(which, U9(*la), U9(*limbs(b)), out)
            v = val(out)
            assert v % mod == a * b * rinv % mod
            assert normalised(out) and v * RP < a * b + mod * RP, (hex(a), hex(b))
            ub = U9()
            h.h_mul29_ub(which, U9(*la), U9(*limbs(b)), ub)      # the wave-uniform-factor form (evaluator constants): same limbs
            assert list(ub) == list(out)
    for a in a_vals:
        if a * a >= RP * mod:
            continue
        out, ref = U9(), U9()
        h.h_sqr29(which, U9(*limbs(a)), out)
        h.h_mul29(which, U9(*limbs(a)), U9(*limbs(a)), ref)
        assert list(out) == list(ref)            # the square is the product, limb for limb


@pytest.mark.parametrize("which", [0, 1])
def test_two_product_pass(h, which):
    """mul2add29(a, b, c, d) = (a b + c d) 2^-261 mod p with d = K p - x taken limb-wise (limbs up to 2^30), columns at their
    maximum (all limbs 2^29 - 1 / 2^30 - 1) included: no 64-bit column may overflow."""
    rng = random.Random(13 + which)
    mod = MOD[which]
    rinv = pow(RP, -1, mod)
    for x in edge_values(mod, rng, 2) + [2 * mod - (1 << 232) - 1]:
        if x >> 232 >= (2 * mod) >> 232:         # the top limb must stay below that of 2p
            continue
        out = U9()
        h.h_neg29k2(which, U9(*limbs(x)), out)
        assert val(out) == 2 * mod - x and all(int(v) < 1 << 30 for v in out)
    full = [M29] * 8
    worst = (U9(*(full + [M29])), U9(*(full + [0x30644e * 2])), U9(*(full + [M29])), U9(*([(1 << 30) - 1] * 8 + [0x30644e * 2])))
    out = U9()
    h.h_mul2add29(which, *worst, out)            # value far beyond the a b + c d < 2^261 p precondition: only the congruence is asked
    a, b, c, d = (val(w) for w in worst)
    assert val(out) % mod == (a * b + c * d) * rinv % mod
    for _ in range(200):
        a, b = rng.randrange(10 * mod), rng.randrange(8 * mod)
        c, x = rng.randrange(8 * mod), rng.randrange(2 * mod)
        if rng.random() < 0.2:
            a, b, c, x = 10 * mod - 1, 8 * mod - 1, 8 * mod - 1, 0
        dl = U9()
        h.h_neg29k2(which, U9(*limbs(x)), dl)
        d = 2 * mod - x
        assert a * b + c * d < RP * mod
        out = U9()
        h.h_mul2add29(which, U9(*limbs(a)), U9(*limbs(b)), U9(*limbs(c)), dl, out)
        v = val(out)
        assert v % mod == (a * b + c * d) * rinv % mod
        assert normalised(out) and v * RP < a * b + c * d + mod * RP


@pytest.mark.parametrize("which", [0, 1])
def test_lazy_reduction_and_inverse(h, which):
    rng = random.Random(11 + which)
    mod = MOD[which]
    around_multiples = [k_ * mod + d for k_ in range(1, 64) for d in (-1, 0, 1)] + [64 * mod - 1]      # the quotient estimate may be one short, never over
    for v in edge_values(mod, rng, 64) + around_multiples + [rng.randrange(64 * mod) for _ in range(300)]:          # reduce_lazy29: normalised lazy value < 64 m
        o8 = U8()
        h.h_reduce_lazy29(which, U9(*limbs(v)), o8)
        assert sum(int(x) << (32 * i) for i, x in enumerate(o8)) == v % mod
    r256 = (1 << 256) % mod
    for x in [1, 2, mod - 1] + [rng.randrange(1, mod) for _ in range(6)]:
        xm = x * r256 % mod                      # R = 2^256 Montgomery form in and out
        o8 = U8()
        h.h_inv_via29(which, U8(*[(xm >> (32 * i)) & 0xFFFFFFFF for i in range(8)]), o8)
        got = sum(int(v) << (32 * i) for i, v in enumerate(o8))
        assert got == pow(x, -1, mod) * r256 % mod


@pytest.mark.parametrize("which", [0, 1])
def test_binary_euclid_inverse(h, which):
    """inv_xgcd (ff.hip.hpp): x^-1 mod m of a plain integer -- what the lone inverting lane of the batch inversion runs instead of
    the 381-product Fermat ladder; 0 maps to 0; the edges: 1, 2, m - 1, (m + 1) / 2, powers of two (long halving runs), values
    one subtraction away from the modulus"""
    rng = random.Random(23 + which)
    mod = MOD[which]
    xs = [1, 2, 3, mod - 1, mod - 2, (mod + 1) // 2, (mod - 1) // 2, 1 << 253, 1 << 128, (1 << 253) + 1, (1 << 200) - 1] + [rng.randrange(1, mod) for _ in range(200)]
    for x in xs:
        o8 = U8()
        h.h_inv_xgcd(which, U8(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)]), o8)
        got = sum(int(v) << (32 * i) for i, v in enumerate(o8))
        assert got == pow(x, -1, mod), hex(x)
    o8 = U8(*([7] * 8))
    h.h_inv_xgcd(which, U8(*([0] * 8)), o8)
    assert list(o8) == [0] * 8


def test_subtractions_keep_limbs_non_negative(h):
    """sub_n<K>(a, b) = a - b + K p for a normalised b < K p: no limb may wrap, the result is normalised."""
    rng = random.Random(3)
    for K in (1, 2, 3, 4, 5, 6, 8):
        for b in edge_values(P, rng, K):
            for a in (0, 1, P - 1, rng.randrange(8 * P), 8 * P - 1):
                out = U9()
                assert h.h_sub_n(K, U9(*limbs(a)), U9(*limbs(b)), out) == 0
                assert val(out) == a - b + K * P and normalised(out)
    for K, S in ((3, 2), (4, 3)):                # subtrahend = uncarried sum of S normalised values, together below K p
        for _ in range(60):
            parts = [rng.randrange(K * P // S) for _ in range(S)]
            if rng.random() < 0.3:
                parts = [K * P // S - 1 - (1 << 232)] * S
            bl = [sum(limbs(v)[i] for v in parts) for i in range(9)]
            if rng.random() < 0.3:               # every limb at its maximum
                bl = [S * M29] * 8 + [bl[8]]
            b = val(bl)
            a = rng.choice([0, 1, rng.randrange(8 * P)])
            out = U9()
            assert h.h_sub_nw(K, S, U9(*limbs(a)), U9(*bl), out) == 0
            assert b < K * P and val(out) == a - b + K * P and normalised(out)
    for maxk in (3, 9):
        for k in range(0, maxk + 1):
            assert h.h_is_zero_mod_p(maxk, U9(*limbs(k * P))) == 1
            assert h.h_is_zero_mod_p(maxk, U9(*limbs(k * P + 1))) == 0
        assert h.h_is_zero_mod_p(maxk, U9(*limbs(rng.randrange(1, P)))) == 0
        assert h.h_is_zero_mod_p(maxk, U9(*limbs((maxk + 1) * P))) == 0          # a multiple of p, but beyond the range asked for
        for k in range(1, maxk + 1):             # same low limbs as k p, another top limb
            assert h.h_is_zero_mod_p(maxk, U9(*(limbs(k * P)[:8] + [limbs(k * P)[8] + 1]))) == 0
            assert h.h_is_zero_mod_p(maxk, U9(*(limbs(k * P)[:4] + [limbs(k * P)[4] ^ 1] + limbs(k * P)[5:]))) == 0


# ---- group law ------------------------------------------------------------------------------------------------------

def to_rp(v):
    return v * RP % P


def affine_limbs(pt):
    if pt is None:
        return [0] * 18
    return limbs(to_rp(pt[0])) + limbs(to_rp(pt[1]))


def xyzz_limbs(pt, rng, lazy=True):
    """a representation (x zz, y zzz, zz, zzz) of pt with a random zz = z^2, zzz = z^3, R' form; x, y < 8p, zz, zzz < 2p"""
    if pt is None:
        return [rng.randrange(1 << 29) for _ in range(18)] + [0] * 9 + [rng.randrange(1 << 29) for _ in range(9)]
    z = rng.randrange(1, P)
    zz, zzz = z * z % P, z * z * z % P
    x, y = pt[0] * zz % P, pt[1] * zzz % P
    k = (lambda m: rng.randrange(m)) if lazy else (lambda m: 0)
    return limbs(to_rp(x) + k(7) * P) + limbs(to_rp(y) + k(7) * P) + limbs(to_rp(zz) + k(1) * P) + limbs(to_rp(zzz) + k(1) * P)


def from_xyzz(l):
    x, y, zz, zzz = (val(l[9 * i: 9 * i + 9]) for i in range(4))
    if zz == 0:
        return None
    rinv = pow(RP, -1, P)
    x, y, zz, zzz = (v * rinv % P for v in (x, y, zz, zzz))
    assert zz % P != 0 and pow(zz, 3, P) == zzz * zzz % P
    return (x * pow(zz, -1, P) % P, y * pow(zzz, -1, P) % P)


def check_stored_invariant(l):
    x, y, zz, zzz = (val(l[9 * i: 9 * i + 9]) for i in range(4))
    for i in range(4):
        assert normalised(l[9 * i: 9 * i + 9])
    assert x < 8 * P and y < 8 * P and zz < 2 * P and zzz < 2 * P


def points(rng, n):
    g = (1, 2)
    return [o.g1_mul(g, rng.randrange(1, R)) for _ in range(n)]


def same(a, b):
    if a is None or b is None:
        return a is None and b is None
    return (a[0] % P, a[1] % P) == (b[0] % P, b[1] % P)


def test_group_law_against_the_oracle(h):
    rng = random.Random(29)
    pts = points(rng, 6)
    cases = [(a, b) for a in pts[:3] for b in pts[3:]]
    cases += [(pts[0], pts[0]), (pts[1], o.g1_neg(pts[1])), (None, pts[2]), (pts[2], None), (None, None)]
    for a, b in cases:
        want = o.g1_add(a, b)
        for _ in range(3):
            pa = xyzz_limbs(a, rng)
            out = U36()
            h.h_madd29(U36(*pa), U18(*affine_limbs(b)), out)     # mixed addition, every exceptional case
            assert same(from_xyzz(list(out)), want)
            if from_xyzz(list(out)) is not None and not (a is None):
                check_stored_invariant(list(out))
            out2 = U36()
            h.h_add29pt(U36(*pa), U36(*xyzz_limbs(b, rng, lazy=False)), out2)
            assert same(from_xyzz(list(out2)), want)
    for a in pts:
        want = o.g1_add(a, a)
        out = U36()
        h.h_dbl29pt(U36(*xyzz_limbs(a, rng)), out)
        assert same(from_xyzz(list(out)), want)
        check_stored_invariant(list(out))
        h.h_dbl_affine29(U18(*affine_limbs(a)), out)
        assert same(from_xyzz(list(out)), want)
        check_stored_invariant(list(out))
    one = U9()
    h.h_one29(one)
    assert val(one) == RP % P


def test_accumulation_chain_stays_within_bounds(h):
    """a bucket's life: identity, then 200 mixed additions of lazily stored results -- the stored invariant must hold at every step"""
    rng = random.Random(31)
    pts = points(rng, 8)
    acc_l = xyzz_limbs(None, rng)
    acc = None
    for i in range(200):
        q = pts[rng.randrange(len(pts))]
        out = U36()
        h.h_madd29(U36(*acc_l), U18(*affine_limbs(q)), out)
        acc_l = list(out)
        acc = o.g1_add(acc, q)
        assert same(from_xyzz(acc_l), acc)
        if acc is not None and i > 0:
            check_stored_invariant(acc_l)


def test_exceptional_cases_inside_a_chain(h):
    """doublings (q = the accumulated point) and cancellations (q = its inverse) met by lazily reduced accumulators in the middle of a
    chain: the k p test of the mixed addition sees differences 8p + (u2 - x1) with every multiple of p that the bounds allow"""
    for seed in range(6):
        rng = random.Random(500 + seed)
        pts = points(rng, 4)
        acc_l, acc = xyzz_limbs(None, rng), None
        for i in range(60):
            r = rng.random()
            q = acc if (acc is not None and r < 0.1) else o.g1_neg(acc) if (acc is not None and r < 0.2) else pts[rng.randrange(len(pts))]
            out = U36()
            h.h_madd29(U36(*acc_l), U18(*affine_limbs(q)), out)
            acc_l, acc = list(out), o.g1_add(acc, q)
            assert same(from_xyzz(acc_l), acc), (seed, i)
            # a lazily reduced representation of the same point goes into the next step
            if acc is not None and rng.random() < 0.5:
                acc_l = xyzz_limbs(acc, rng)
