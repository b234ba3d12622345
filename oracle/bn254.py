"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- BN254 field / curve / NTT / MSM in Python big-ints.

This file is a *restatement* of the arithmetic that the reference reaches through its external
dependencies; it is the checker for the HIP path, never the product path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Where the algorithm lives (none of it is under /root/reference -- see SURVEY.md section 8c):
  * halo2curves 0.1.0 @ scroll-tech/halo2curves a495a7b  (``bn256::{Fr,Fq,G1Affine}``)
      [REF Cargo.lock:2239-2241]
  * halo2_proofs 1.1.0 @ scroll-tech/halo2 e5ddf67 (``arithmetic::{best_fft,best_multiexp}``,
      ``poly::EvaluationDomain``) [REF Cargo.lock:2214-2216]
The published algorithms are restated here from their mathematical definitions.

Pinning ("parity pinned" items -- checked in tests/test_oracle_golden.py):
  G3  third MockProver challenge, pins ``Fr::from_uniform_bytes`` (64-byte LE mod r) and r
        [REF zkevm-circuits/src/super_circuit.rs:729]
  G4  Fq modulus - 2 as an EVM word        [REF zkevm-circuits/src/ecc_circuit/test.rs:208]
  G5  ecAdd((1,2),(1,2)) == ecMul((1,2),2) == (0x0306..cfd3, 0x15ed..a2c4)
        [REF bus-mapping/src/evm/opcodes/callop.rs:883-917]
  G6  ecPairing check vector (two pairs, result 1) pins G2 / Fq2 / Fq12 tower + ate pairing
        [REF bus-mapping/src/evm/opcodes/callop.rs:925-936]
Raw NTT / MSM outputs have NO known-answer vectors in the reference ("parity unpinned by the
reference" for those: they are pinned by mathematical definition -- naive O(n^2) DFT and naive
double-and-add sum -- which this module also provides).
"""
from __future__ import annotations

import hashlib
from typing import Iterable, List, Optional, Sequence, Tuple

# ----------------------------------------------------------------------------------------------
# Constants (arithmetic identities re-checked in tests/test_oracle_golden.py)
# ----------------------------------------------------------------------------------------------
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # Fr modulus r
P_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # Fq modulus p
FR_S = 28                      # two-adicity of r-1
FR_GENERATOR = 7               # multiplicative generator used by halo2curves Fr
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (R_MOD - 1) >> FR_S, R_MOD)   # primitive 2^28-th root
FR_DELTA = pow(FR_GENERATOR, 1 << FR_S, R_MOD)                     # generator of the t-order subgroup
FR_ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23  # cube root of unity
MONT_R_BITS = 256
FR_MONT_R = (1 << MONT_R_BITS) % R_MOD
FQ_MONT_R = (1 << MONT_R_BITS) % P_MOD
FR_MONT_R2 = (FR_MONT_R * FR_MONT_R) % R_MOD
FQ_MONT_R2 = (FQ_MONT_R * FQ_MONT_R) % P_MOD
FR_INV64 = (-pow(R_MOD, -1, 1 << 64)) % (1 << 64)
FQ_INV64 = (-pow(P_MOD, -1, 1 << 64)) % (1 << 64)
FR_INV32 = FR_INV64 & 0xFFFFFFFF
FQ_INV32 = FQ_INV64 & 0xFFFFFFFF
CURVE_B = 3
G1_GEN = (1, 2)


# ----------------------------------------------------------------------------------------------
# Field helpers
# ----------------------------------------------------------------------------------------------
def fr_inv(a: int) -> int:
    return pow(a, -1, R_MOD)


def fq_inv(a: int) -> int:
    return pow(a, -1, P_MOD)


def fr_from_uniform_bytes(b: bytes) -> int:
    """``Fr::from_uniform_bytes``: 64 little-endian bytes reduced mod r (halo2curves bn256/fr.rs)."""
    assert len(b) == 64
    return int.from_bytes(b, "little") % R_MOD


def to_mont(a: int, mod: int) -> int:
    return (a << MONT_R_BITS) % mod


def from_mont(a: int, mod: int) -> int:
    return (a * pow(1 << MONT_R_BITS, -1, mod)) % mod


def limbs_le(a: int, n: int = 4, bits: int = 64) -> List[int]:
    mask = (1 << bits) - 1
    return [(a >> (bits * i)) & mask for i in range(n)]


def mont_bytes(a: int, mod: int) -> bytes:
    """In-memory / ``SerdeFormat::RawBytes`` encoding: 4xu64 LE limbs of a*R mod m."""
    return to_mont(a, mod).to_bytes(32, "little")


def from_mont_bytes(b: bytes, mod: int) -> int:
    return from_mont(int.from_bytes(b, "little"), mod)


def mont_mul_cios(a: int, b: int, mod: int, inv64: int) -> int:
    """Word-level CIOS Montgomery product on 4x64-bit limbs -- restates what halo2curves'
    ``field_arithmetic!``/``montgomery_reduce`` computes: a*b*R^-1 mod m for a,b < m.
    Kept word-level so the C and HIP limbs code can be diffed against it step by step."""
    A = limbs_le(a)
    B = limbs_le(b)
    M = limbs_le(mod)
    W = 1 << 64
    t = [0] * 6
    for i in range(4):
        c = 0
        for j in range(4):
            s = t[j] + A[j] * B[i] + c
            t[j] = s % W
            c = s // W
        s = t[4] + c
        t[4] = s % W
        t[5] = s // W
        m = (t[0] * inv64) % W
        s = t[0] + m * M[0]
        c = s // W
        for j in range(1, 4):
            s = t[j] + m * M[j] + c
            t[j - 1] = s % W
            c = s // W
        s = t[4] + c
        t[3] = s % W
        t[4] = t[5] + s // W
    r = sum(t[i] << (64 * i) for i in range(5))
    if r >= mod:
        r -= mod
    return r


# ----------------------------------------------------------------------------------------------
# G1 (y^2 = x^3 + 3 over Fq).  Affine points are (x, y) tuples; identity is None.
# In-memory halo2curves encoding of the identity is (0, 0)  [EXT-RECALL, SURVEY B.1].
# ----------------------------------------------------------------------------------------------
Affine = Optional[Tuple[int, int]]
Jac = Tuple[int, int, int]  # (X, Y, Z), identity has Z == 0


def g1_is_on_curve(pt: Affine) -> bool:
    if pt is None:
        return True
    x, y = pt
    if not (0 <= x < P_MOD and 0 <= y < P_MOD):
        return False
    return (y * y - x * x * x - CURVE_B) % P_MOD == 0


def g1_neg(pt: Affine) -> Affine:
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P_MOD)


def g1_add(p1: Affine, p2: Affine) -> Affine:
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P_MOD == 0:
            return None
        lam = (3 * x1 * x1) * fq_inv(2 * y1) % P_MOD
    else:
        lam = (y2 - y1) * fq_inv((x2 - x1) % P_MOD) % P_MOD
    x3 = (lam * lam - x1 - x2) % P_MOD
    y3 = (lam * (x1 - x3) - y1) % P_MOD
    return (x3, y3)


JAC_ID: Jac = (1, 1, 0)


def jac_from_affine(p: Affine) -> Jac:
    return JAC_ID if p is None else (p[0], p[1], 1)


def jac_to_affine(p: Jac) -> Affine:
    X, Y, Z = p
    if Z % P_MOD == 0:
        return None
    zi = fq_inv(Z)
    zi2 = zi * zi % P_MOD
    return (X * zi2 % P_MOD, Y * zi2 * zi % P_MOD)


def jac_double(p: Jac) -> Jac:
    X, Y, Z = p
    if Z == 0 or Y == 0:
        return JAC_ID
    A = X * X % P_MOD
    B = Y * Y % P_MOD
    C = B * B % P_MOD
    D = 2 * ((X + B) * (X + B) - A - C) % P_MOD
    E = 3 * A % P_MOD
    F = E * E % P_MOD
    X3 = (F - 2 * D) % P_MOD
    Y3 = (E * (D - X3) - 8 * C) % P_MOD
    Z3 = 2 * Y * Z % P_MOD
    return (X3, Y3, Z3)


def jac_add(p: Jac, q: Jac) -> Jac:
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    if Z1 == 0:
        return q
    if Z2 == 0:
        return p
    Z1Z1 = Z1 * Z1 % P_MOD
    Z2Z2 = Z2 * Z2 % P_MOD
    U1 = X1 * Z2Z2 % P_MOD
    U2 = X2 * Z1Z1 % P_MOD
    S1 = Y1 * Z2 * Z2Z2 % P_MOD
    S2 = Y2 * Z1 * Z1Z1 % P_MOD
    if U1 == U2:
        if S1 == S2:
            return jac_double(p)
        return JAC_ID
    H = (U2 - U1) % P_MOD
    I = 4 * H * H % P_MOD
    J = H * I % P_MOD
    rr = 2 * (S2 - S1) % P_MOD
    V = U1 * I % P_MOD
    X3 = (rr * rr - J - 2 * V) % P_MOD
    Y3 = (rr * (V - X3) - 2 * S1 * J) % P_MOD
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) * H % P_MOD
    return (X3, Y3, Z3)


def jac_add_affine(p: Jac, q: Affine) -> Jac:
    return jac_add(p, jac_from_affine(q))


def g1_mul(pt: Affine, k: int) -> Affine:
    k %= R_MOD
    acc = JAC_ID
    base = jac_from_affine(pt)
    while k:
        if k & 1:
            acc = jac_add(acc, base)
        base = jac_double(base)
        k >>= 1
    return jac_to_affine(acc)


def g1_affine_bytes_raw(pt: Affine) -> bytes:
    """``SerdeFormat::RawBytes`` / in-memory image of ``G1Affine``: x||y, Montgomery limbs, LE.
    Identity is (0,0)."""
    if pt is None:
        return bytes(64)
    return mont_bytes(pt[0], P_MOD) + mont_bytes(pt[1], P_MOD)


def g1_affine_from_bytes_raw(b: bytes) -> Affine:
    assert len(b) == 64
    if b == bytes(64):
        return None
    return (from_mont_bytes(b[:32], P_MOD), from_mont_bytes(b[32:], P_MOD))


def g1_compress(pt: Affine) -> bytes:
    """Proof-byte / `SerdeFormat::Processed` encoding of halo2curves @ a495a7b (`new_curve_impl!`, flags in the two spare
    top bits of the last byte): x canonical LE with the parity of y in bit 254 (0x40 of byte 31); the identity is the
    zero x with bit 255 (0x80) set.  PINNED by the reference's own data: all 7 vk points and all 11 proof points of
    aggregator/data/batch-task.json carry (y & 1) << 6 and never bit 7 (tests/test_reference_chunk_proof.py).  The
    identity's image cannot be read off that fixture (no identity in it): it is the encoding the same macro writes
    [EXT-RECALL derive/curve.rs `to_bytes`]."""
    if pt is None:
        return bytes(31) + b"\x80"
    x, y = pt
    b = bytearray(x.to_bytes(32, "little"))
    b[31] |= (y & 1) << 6
    return bytes(b)


def g1_decompress(raw: bytes) -> Affine:
    """inverse of g1_compress (`from_bytes`): bit 255 = identity flag (then x and the parity flag must be zero), bit 254 = parity
    of y; x must be canonical; raises ValueError where halo2curves returns `CtOption::none`"""
    if len(raw) != 32:
        raise ValueError("compressed G1 is 32 bytes")
    v = int.from_bytes(raw, "little")
    is_inf, sign, x = v >> 255, (v >> 254) & 1, v & ((1 << 254) - 1)
    if is_inf:
        if x != 0 or sign:
            raise ValueError("identity flag on a non-zero encoding")
        return None
    if x >= P_MOD:
        raise ValueError("non-canonical x")
    y2 = (x * x * x + CURVE_B) % P_MOD
    y = pow(y2, (P_MOD + 1) // 4, P_MOD)
    if y * y % P_MOD != y2:
        raise ValueError("x is not the abscissa of a curve point")
    if (y & 1) != sign:
        y = P_MOD - y
    return (x, y)


# ----------------------------------------------------------------------------------------------
# MSM
# ----------------------------------------------------------------------------------------------
def msm_naive(scalars: Sequence[int], bases: Sequence[Affine]) -> Affine:
    """Definition: sum_i s_i * P_i (double-and-add per term)."""
    acc = JAC_ID
    for s, b in zip(scalars, bases):
        t = g1_mul(b, s)
        acc = jac_add_affine(acc, t)
    return jac_to_affine(acc)


def msm_pippenger_halo2(scalars: Sequence[int], bases: Sequence[Affine]) -> Affine:
    """Restates halo2_proofs ``arithmetic::multiexp_serial`` (unsigned c-bit windows,
    c = ceil(ln n) for n >= 32, 3 for n >= 4, else 1; (256/c)+1 segments processed from the top
    with c doublings between, buckets summed by running sum)."""
    import math

    n = len(bases)
    if n < 4:
        c = 1
    elif n < 32:
        c = 3
    else:
        c = int(math.ceil(math.log(n)))
    segments = 256 // c + 1
    acc = JAC_ID
    for seg in reversed(range(segments)):
        for _ in range(c):
            acc = jac_double(acc)
        buckets: List[Jac] = [JAC_ID] * ((1 << c) - 1)
        for s, b in zip(scalars, bases):
            d = (s >> (seg * c)) & ((1 << c) - 1)
            if d:
                buckets[d - 1] = jac_add_affine(buckets[d - 1], b)
        running = JAC_ID
        for bk in reversed(buckets):
            running = jac_add(bk, running)
            acc = jac_add(acc, running)
    return jac_to_affine(acc)


# ----------------------------------------------------------------------------------------------
# NTT over Fr
# ----------------------------------------------------------------------------------------------
def omega_for_k(k: int) -> int:
    """``EvaluationDomain``: omega = ROOT_OF_UNITY^(2^(S-k))."""
    assert 0 <= k <= FR_S
    return pow(FR_ROOT_OF_UNITY, 1 << (FR_S - k), R_MOD)


def ntt_naive(a: Sequence[int], omega: int) -> List[int]:
    """Definition: out[i] = sum_j a[j] * omega^(i*j)."""
    n = len(a)
    out = []
    for i in range(n):
        w = pow(omega, i, R_MOD)
        acc = 0
        x = 1
        for j in range(n):
            acc = (acc + a[j] * x) % R_MOD
            x = x * w % R_MOD
        out.append(acc)
    return out


def bitrev(i: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def best_fft(a: List[int], omega: int, log_n: int) -> None:
    """Restates halo2_proofs ``arithmetic::best_fft`` (serial branch): bit-reversal permutation,
    then radix-2 DIT stages with the twiddle table ``twiddles[i] = omega^i``.  In place,
    natural order in -> natural order out."""
    n = 1 << log_n
    assert len(a) == n
    for k in range(n):
        rk = bitrev(k, log_n)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    tw = [1] * (n // 2 if n > 1 else 1)
    for i in range(1, n // 2):
        tw[i] = tw[i - 1] * omega % R_MOD
    chunk = 2
    tchunk = n // 2
    for _ in range(log_n):
        half = chunk // 2
        for base in range(0, n, chunk):
            for i in range(half):
                t = a[base + half + i] * tw[i * tchunk] % R_MOD
                u = a[base + i]
                a[base + i] = (u + t) % R_MOD
                a[base + half + i] = (u - t) % R_MOD
        chunk *= 2
        tchunk //= 2


def ifft(a: List[int], log_n: int) -> None:
    """``EvaluationDomain::lagrange_to_coeff``: FFT with omega^-1, then scale by n^-1."""
    om_inv = fr_inv(omega_for_k(log_n))
    best_fft(a, om_inv, log_n)
    ninv = fr_inv(1 << log_n)
    for i in range(len(a)):
        a[i] = a[i] * ninv % R_MOD


def eval_polynomial(coeffs: Sequence[int], x: int) -> int:
    """``arithmetic::eval_polynomial`` (Horner)."""
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R_MOD
    return acc


def kate_division(coeffs: Sequence[int], z: int) -> List[int]:
    """``arithmetic::kate_division``: quotient of (f(X) - f(z)) / (X - z); len n-1."""
    n = len(coeffs)
    q = [0] * (n - 1)
    tmp = 0
    for i in reversed(range(n - 1)):
        tmp = (coeffs[i + 1] + tmp * z) % R_MOD
        q[i] = tmp
    return q


class EvaluationDomain:
    """Restates halo2_proofs ``poly::EvaluationDomain::new(j, k)`` and its transforms
    (SURVEY Appendix B.2).  g_coset = ZETA; the extended coset is zeta * <extended_omega>."""

    def __init__(self, j: int, k: int):
        self.k = k
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k = ek
        self.extended_omega = omega_for_k(ek)
        self.omega = pow(self.extended_omega, 1 << (ek - k), R_MOD)
        assert self.omega == omega_for_k(k)
        self.omega_inv = fr_inv(self.omega)
        self.extended_omega_inv = fr_inv(self.extended_omega)
        self.g_coset = FR_ZETA
        self.g_coset_inv = FR_ZETA * FR_ZETA % R_MOD
        self.ifft_divisor = fr_inv(1 << k)
        self.extended_ifft_divisor = fr_inv(1 << ek)
        # t_evaluations: 1/((zeta*extended_omega^i)^n - 1), period 2^(ek-k)
        t = []
        cur = pow(self.g_coset, self.n, R_MOD)
        orig = cur
        step = pow(self.extended_omega, self.n, R_MOD)
        while True:
            t.append(cur)
            cur = cur * step % R_MOD
            if cur == orig:
                break
        self.t_evaluations = [fr_inv((v - 1) % R_MOD) for v in t]

    def lagrange_to_coeff(self, a: Sequence[int]) -> List[int]:
        a = list(a)
        best_fft(a, self.omega_inv, self.k)
        return [v * self.ifft_divisor % R_MOD for v in a]

    def coeff_to_lagrange(self, a: Sequence[int]) -> List[int]:
        a = list(a)
        best_fft(a, self.omega, self.k)
        return a

    def coeff_to_extended(self, a: Sequence[int]) -> List[int]:
        a = list(a)
        assert len(a) == self.n
        z = 1
        for i in range(len(a)):          # distribute_powers_zeta: a[i] *= zeta^i
            a[i] = a[i] * z % R_MOD
            z = z * self.g_coset % R_MOD
        a += [0] * ((1 << self.extended_k) - self.n)
        best_fft(a, self.extended_omega, self.extended_k)
        return a

    def extended_to_coeff(self, a: Sequence[int]) -> List[int]:
        a = list(a)
        assert len(a) == 1 << self.extended_k
        best_fft(a, self.extended_omega_inv, self.extended_k)
        z = 1
        for i in range(len(a)):
            a[i] = a[i] * self.extended_ifft_divisor % R_MOD * z % R_MOD
            z = z * self.g_coset_inv % R_MOD
        return a[: self.n * self.quotient_poly_degree]

    def rotate_omega(self, x: int, rot: int) -> int:
        return x * pow(self.omega, rot % self.n, R_MOD) % R_MOD


# ----------------------------------------------------------------------------------------------
# Deterministic test-input generators shared by oracle, tests and bench.
# splitmix64 stream -> 512 bits -> mod r : same code exists in C (oracle/c) and in the HIP host
# library so that every side can regenerate identical inputs from a seed.
# ----------------------------------------------------------------------------------------------
MASK64 = (1 << 64) - 1


def splitmix64(state: int) -> Tuple[int, int]:
    state = (state + 0x9E3779B97F4A7C15) & MASK64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return state, z ^ (z >> 31)


def rand_fr_stream(seed: int, n: int) -> List[int]:
    """n field elements: element i = (4 consecutive splitmix64 words, LE, top two bits cleared)
    reduced once by conditional subtraction -- cheap enough to regenerate 2^20+ values in C."""
    out = []
    st = seed & MASK64
    for _ in range(n):
        v = 0
        for w in range(4):
            st, z = splitmix64(st)
            v |= z << (64 * w)
        v &= (1 << 254) - 1
        if v >= R_MOD:
            v -= R_MOD
        out.append(v)
    return out


def srs_powers(s: int, n: int) -> List[Affine]:
    """``ParamsKZG::unsafe_setup_with_s``: g[i] = s^i * G1."""
    out = []
    cur = 1
    for _ in range(n):
        out.append(g1_mul(G1_GEN, cur))
        cur = cur * s % R_MOD
    return out


def mock_prover_challenge(i: int) -> int:
    """halo2 ``dev.rs``: h0 = blake2b-512("Halo2-MockProver"); h_i = blake2b-512(h_{i-1});
    challenge_i = from_uniform_bytes(h_i).  (golden G3)."""
    h = hashlib.blake2b(b"Halo2-MockProver", digest_size=64).digest()
    for _ in range(i):
        h = hashlib.blake2b(h, digest_size=64).digest()
    return fr_from_uniform_bytes(h)


# ---- ChaCha20 counter-mode field sampling (zk_fr_random) ------------------------------------------
# Restates RFC 7539 section 2.3 (block function) with the original 64-bit counter / 64-bit stream
# layout (state words 12..13 = block counter, 14..15 = stream id), followed by halo2curves'
# Fr::from_uniform_bytes (lo + hi * 2^256 mod r over 64 little-endian bytes).
def chacha20_block(key32: bytes, counter: int, stream: int) -> bytes:
    import struct
    k = struct.unpack("<8I", key32)
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574, *k,
            counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF, stream & 0xFFFFFFFF, (stream >> 32) & 0xFFFFFFFF]
    x = list(init)

    def rotl(v, c):
        return ((v << c) & 0xFFFFFFFF) | (v >> (32 - c))

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return struct.pack("<16I", *[(a + b) & 0xFFFFFFFF for a, b in zip(x, init)])


def fr_random_chacha(key32: bytes, stream: int, first_block: int, n: int):
    return [fr_from_uniform_bytes(chacha20_block(key32, first_block + i, stream)) for i in range(n)]
