"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- BN254 G2 and the optimal-ate pairing in Python.

The reference reaches pairings through ``halo2curves::bn256::Bn256::pairing`` (external crate,
not on disk).  The KZG *verifier* half of the path stays on the CPU in the
reference [REF circuit-benchmarks/src/super_circuit.rs:141-154]; this module exists so the
tests can run a real pairing-based acceptance check on proofs produced by the HIP prover.

Layout and names (`FQP` polynomial-quotient fields, `twist`, `cast_g1_to_fq12`, `linefunc`, `ATE_LOOP_COUNT`, the modulus
coefficients 82 / -18) follow the public py_ecc `bn128` module (Ethereum Foundation, MIT licence), written out again here from
its structure; the mathematics is the textbook construction: Fq2 = Fq[u]/(u^2+1); Fq12 = Fq[w]/(w^12-18w^6+82);
D-type sextic twist with xi = 9+u; Miller loop over 6t+2 = 29793968203157093288 followed by
the two Frobenius line steps; final exponentiation (p^12-1)/r done naively.

Pinned by golden G6: the ecPairing call-data vector in
[REF bus-mapping/src/evm/opcodes/callop.rs:925-936] must evaluate to 1.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

from .bn254 import P_MOD, R_MOD, fq_inv

ATE_LOOP_COUNT = 29793968203157093288
LOG_ATE_LOOP_COUNT = 63
FQ12_MOD_COEFFS = [82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0]  # w^12 = 18 w^6 - 82


class FQP:
    """Element of Fq[w]/(modulus); coefficient list, low degree first."""

    __slots__ = ("c",)
    degree = 0
    mc: Sequence[int] = ()

    def __init__(self, coeffs):
        self.c = [x % P_MOD for x in coeffs]

    @classmethod
    def one(cls):
        return cls([1] + [0] * (cls.degree - 1))

    @classmethod
    def zero(cls):
        return cls([0] * cls.degree)

    def __add__(self, o):
        return type(self)([a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return type(self)([a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return type(self)([-a for a in self.c])

    def __eq__(self, o):
        return self.c == o.c

    def scale(self, k: int):
        return type(self)([a * k for a in self.c])

    def __mul__(self, o):
        if isinstance(o, int):
            return self.scale(o)
        d = self.degree
        b = [0] * (2 * d - 1)
        for i, x in enumerate(self.c):
            if x:
                for j, y in enumerate(o.c):
                    b[i + j] += x * y
        mc = self.mc
        for exp in range(2 * d - 2, d - 1, -1):
            top = b[exp]
            if top:
                for i in range(d):
                    if mc[i]:
                        b[exp - d + i] -= top * mc[i]
        return type(self)(b[:d])

    def __pow__(self, e: int):
        result = type(self).one()
        base = self
        while e:
            if e & 1:
                result = result * base
            base = base * base
            e >>= 1
        return result

    def inv(self):
        # extended Euclid over Fq[w]
        d = self.degree
        lm, hm = [1] + [0] * d, [0] * (d + 1)
        low, high = self.c + [0], list(self.mc) + [1]
        low = [x % P_MOD for x in low]
        high = [x % P_MOD for x in high]

        def deg(p):
            k = len(p) - 1
            while k and p[k] == 0:
                k -= 1
            return k

        def poly_rounded_div(a, b):
            dega, degb = deg(a), deg(b)
            temp = list(a)
            o = [0] * len(a)
            for i in range(dega - degb, -1, -1):
                q = temp[degb + i] * fq_inv(b[degb]) % P_MOD
                o[i] = (o[i] + q) % P_MOD
                for c in range(degb + 1):
                    temp[c + i] = (temp[c + i] - q * b[c]) % P_MOD
            return o[: deg(o) + 1]

        while deg(low):
            r = poly_rounded_div(high, low)
            r += [0] * (d + 1 - len(r))
            nm, new = list(hm), list(high)
            for i in range(d + 1):
                for j in range(d + 1 - i):
                    nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P_MOD
                    new[i + j] = (new[i + j] - low[i] * r[j]) % P_MOD
            lm, low, hm, high = nm, new, lm, low
        li = fq_inv(low[0])
        return type(self)([x * li for x in lm[:d]])


class FQ2(FQP):
    degree = 2
    mc = (1, 0)


class FQ12(FQP):
    degree = 12
    mc = tuple(FQ12_MOD_COEFFS)


G2_GEN = (
    FQ2([
        10857046999023057135944570762232829481370756359578518086990519993285655852781,
        11559732032986387107991004021392285783925812861821192530917403151452391805634,
    ]),
    FQ2([
        8495653923123431417604973247489272438418190587263600148770280649306958101930,
        4082367875863433681332203403145435568316851327593401208105741076214120093531,
    ]),
)
B2 = FQ2([3, 0]) * FQ2([9, 1]).inv()       # twist curve: y^2 = x^3 + 3/(9+u)
B12 = FQ12([3] + [0] * 11)


def is_on_curve(pt, b) -> bool:
    if pt is None:
        return True
    x, y = pt
    return y * y - x * x * x == b


def ec_double(pt):
    if pt is None:
        return None
    x, y = pt
    m = (x * x).scale(3) * (y.scale(2)).inv()
    nx = m * m - x.scale(2)
    ny = m * (x - nx) - y
    return (nx, ny)


def ec_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if y1 == y2:
            return ec_double(p1)
        return None
    m = (y2 - y1) * (x2 - x1).inv()
    nx = m * m - x1 - x2
    ny = m * (x1 - nx) - y1
    return (nx, ny)


def ec_mul(pt, k: int):
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, pt)
        pt = ec_double(pt)
        k >>= 1
    return acc


def ec_neg(pt):
    return None if pt is None else (pt[0], -pt[1])


W = FQ12([0, 1] + [0] * 10)
W2 = W * W
W3 = W2 * W


def twist(pt):
    if pt is None:
        return None
    x, y = pt
    xc = [x.c[0] - x.c[1] * 9, x.c[1]]
    yc = [y.c[0] - y.c[1] * 9, y.c[1]]
    nx = FQ12([xc[0]] + [0] * 5 + [xc[1]] + [0] * 5)
    ny = FQ12([yc[0]] + [0] * 5 + [yc[1]] + [0] * 5)
    return (nx * W2, ny * W3)


def cast_g1_to_fq12(pt):
    if pt is None:
        return None
    x, y = pt
    return (FQ12([x] + [0] * 11), FQ12([y] + [0] * 11))


def linefunc(P1, P2, T):
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if x1 != x2:
        m = (y2 - y1) * (x2 - x1).inv()
        return m * (xt - x1) - (yt - y1)
    if y1 == y2:
        m = (x1 * x1).scale(3) * (y1.scale(2)).inv()
        return m * (xt - x1) - (yt - y1)
    return xt - x1


def miller_loop(Q, P) -> FQ12:
    """Q: twisted G2 point in Fq12 coords, P: G1 point cast into Fq12.  No final exponentiation."""
    if Q is None or P is None:
        return FQ12.one()
    R = Q
    f = FQ12.one()
    for i in range(LOG_ATE_LOOP_COUNT, -1, -1):
        f = f * f * linefunc(R, R, P)
        R = ec_double(R)
        if ATE_LOOP_COUNT & (1 << i):
            f = f * linefunc(R, Q, P)
            R = ec_add(R, Q)
    Q1 = (Q[0] ** P_MOD, Q[1] ** P_MOD)
    nQ2 = (Q1[0] ** P_MOD, -(Q1[1] ** P_MOD))
    f = f * linefunc(R, Q1, P)
    R = ec_add(R, Q1)
    f = f * linefunc(R, nQ2, P)
    return f


def final_exponentiate(f: FQ12) -> FQ12:
    return f ** ((P_MOD ** 12 - 1) // R_MOD)


def pairing(Q, P) -> FQ12:
    """e(P, Q) with P in G1 (affine int tuple or None), Q in G2 (FQ2 tuple or None)."""
    return final_exponentiate(miller_loop(twist(Q), cast_g1_to_fq12(P)))


def pairing_check(pairs) -> bool:
    """prod_i e(P_i, Q_i) == 1 -- one shared final exponentiation, as every KZG verifier does."""
    f = FQ12.one()
    for P, Q in pairs:
        f = f * miller_loop(twist(Q), cast_g1_to_fq12(P))
    return final_exponentiate(f) == FQ12.one()
