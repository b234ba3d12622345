// Host-side BN254 pairing and KZG accumulator arithmetic: between two GPU proofs
// the aggregation layers run `extract_accumulators_and_proof` on the CPU -- random-linear-combine the
// child snarks' KZG accumulators over a Poseidon transcript, check e(lhs, g2) == e(rhs, s_g2), and
// hand the result to the next circuit as 4 x 3 limbs of 88 bits [REF aggregator/src/core.rs:48-147],
// [REF aggregator/src/constants.rs:77-82].  Product code (not the test oracle); the pairing itself
// lives in halo2curves (`Bn256::pairing`, external crate) and is restated here from the textbook
// construction: Fq2 = Fq[u]/(u^2+1), Fq12 = Fq[w]/(w^12 - 18 w^6 + 82) with u -> w^6 - 9, D-type sextic
// twist by xi = 9 + u, Miller loop over 6t + 2 followed by the two Frobenius steps, final
// exponentiation (p^12 - 1)/r.  G2 arithmetic stays in Fq2 (affine, one inversion per step) and
// the line values are assembled directly as sparse Fq12 elements.
// Pinned in tests/ by the reference-held ecPairing vector [REF bus-mapping/src/evm/opcodes/callop.rs:925-936].
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "host_hash.hpp"

namespace zk {
namespace host {

typedef FqC QC;
inline F4 q_add(const F4& a, const F4& b) { return fadd<QC>(a, b); }
inline F4 q_sub(const F4& a, const F4& b) { return fsub<QC>(a, b); }
inline F4 q_mul(const F4& a, const F4& b) { return fmul<QC>(a, b); }
inline F4 q_zero() { F4 z; memset(&z, 0, sizeof z); return z; }
inline F4 q_one() { return fone<QC>(); }
inline F4 q_neg(const F4& a) { return q_sub(q_zero(), a); }
inline F4 q_from_u64(uint64_t v) {          // small integer -> Montgomery form (v * R mod p by repeated doubling of R)
    F4 r = q_zero(), b = q_one();
    while (v) { if (v & 1) r = q_add(r, b); b = q_add(b, b); v >>= 1; }
    return r;
}

// ------------------------------------------------------------------------------------------ Fq2
struct Fq2 { F4 c0, c1; };
inline Fq2 f2_add(const Fq2& a, const Fq2& b) { return {q_add(a.c0, b.c0), q_add(a.c1, b.c1)}; }
inline Fq2 f2_sub(const Fq2& a, const Fq2& b) { return {q_sub(a.c0, b.c0), q_sub(a.c1, b.c1)}; }
inline Fq2 f2_neg(const Fq2& a) { return {q_neg(a.c0), q_neg(a.c1)}; }
inline Fq2 f2_mul(const Fq2& a, const Fq2& b) {
    const F4 t0 = q_mul(a.c0, b.c0), t1 = q_mul(a.c1, b.c1);
    return {q_sub(t0, t1), q_sub(q_sub(q_mul(q_add(a.c0, a.c1), q_add(b.c0, b.c1)), t0), t1)};
}
inline Fq2 f2_scale(const Fq2& a, const F4& k) { return {q_mul(a.c0, k), q_mul(a.c1, k)}; }
inline Fq2 f2_conj(const Fq2& a) { return {a.c0, q_neg(a.c1)}; }
inline Fq2 f2_inv(const Fq2& a) {
    const F4 n = finv<QC>(q_add(q_mul(a.c0, a.c0), q_mul(a.c1, a.c1)));
    return {q_mul(a.c0, n), q_neg(q_mul(a.c1, n))};
}
inline bool f2_is_zero(const Fq2& a) { return fzero<QC>(a.c0) && fzero<QC>(a.c1); }
inline bool f2_eq(const Fq2& a, const Fq2& b) { return memcmp(&a, &b, sizeof a) == 0; }
inline Fq2 f2_one() { return {q_one(), q_zero()}; }
// a^e for a little-endian multi-limb exponent
inline Fq2 f2_pow(Fq2 b, const uint64_t* e, int limbs) {
    Fq2 r = f2_one();
    for (int i = 0; i < 64 * limbs; ++i) { if ((e[i >> 6] >> (i & 63)) & 1) r = f2_mul(r, b); b = f2_mul(b, b); }
    return r;
}

struct G2Affine { Fq2 x, y; };                         // 128 bytes, Montgomery limbs; identity = all zero (halo2curves layout)
inline bool g2_is_identity(const G2Affine& p) { return f2_is_zero(p.x) && f2_is_zero(p.y); }

// ------------------------------------------------------------------------------------------ Fq12
struct Fq12 { F4 c[12]; };
inline Fq12 f12_one() { Fq12 r; for (F4& v : r.c) v = q_zero(); r.c[0] = q_one(); return r; }
inline bool f12_eq(const Fq12& a, const Fq12& b) { return memcmp(&a, &b, sizeof a) == 0; }
inline Fq12 f12_mul(const Fq12& a, const Fq12& b) {
    static const F4 k18 = q_from_u64(18), k82 = q_from_u64(82);
    F4 t[23];
    for (F4& v : t) v = q_zero();
    for (int i = 0; i < 12; ++i) {
        if (fzero<QC>(a.c[i])) continue;                   // line values are sparse
        for (int j = 0; j < 12; ++j) {
            if (fzero<QC>(b.c[j])) continue;
            t[i + j] = q_add(t[i + j], q_mul(a.c[i], b.c[j]));
        }
    }
    for (int i = 22; i >= 12; --i) {                       // w^12 = 18 w^6 - 82
        if (fzero<QC>(t[i])) continue;
        t[i - 6] = q_add(t[i - 6], q_mul(t[i], k18));
        t[i - 12] = q_sub(t[i - 12], q_mul(t[i], k82));
    }
    Fq12 r;
    memcpy(r.c, t, sizeof r.c);
    return r;
}
inline Fq12 f12_pow(Fq12 b, const uint64_t* e, int limbs) {
    Fq12 r = f12_one();
    for (int i = 0; i < 64 * limbs; ++i) { if ((e[i >> 6] >> (i & 63)) & 1) r = f12_mul(r, b); b = f12_mul(b, b); }
    return r;
}
// a + b u  ->  (a - 9 b) + b w^6, placed at w^shift
inline void f12_place(Fq12* r, const Fq2& v, int shift) {
    static const F4 k9 = q_from_u64(9);
    r->c[shift] = q_add(r->c[shift], q_sub(v.c0, q_mul(v.c1, k9)));
    r->c[shift + 6] = q_add(r->c[shift + 6], v.c1);
}

// ------------------------------------------------------------------------------------------ pairing
struct PairingConsts {
    Fq2 frob_x, frob_y;             // xi^((p-1)/3), xi^((p-1)/2): Frobenius of a twisted point
    std::vector<uint64_t> final_exp;    // (p^12 - 1) / r, little-endian limbs
    PairingConsts();
};
// multi-precision helpers for the one-off constants (little-endian u64 limbs)
inline std::vector<uint64_t> mp_mul(const std::vector<uint64_t>& a, const std::vector<uint64_t>& b) {
    std::vector<uint64_t> r(a.size() + b.size(), 0);
    for (size_t i = 0; i < a.size(); ++i) {
        u128 c = 0;
        for (size_t j = 0; j < b.size(); ++j) { c += (u128)a[i] * b[j] + r[i + j]; r[i + j] = (uint64_t)c; c >>= 64; }
        r[i + b.size()] += (uint64_t)c;
    }
    while (r.size() > 1 && r.back() == 0) r.pop_back();
    return r;
}
inline std::vector<uint64_t> mp_sub_small(std::vector<uint64_t> a, uint64_t v) {
    for (size_t i = 0; i < a.size() && v; ++i) { const uint64_t o = a[i]; a[i] -= v; v = o < v ? 1 : 0; }
    return a;
}
// a / d for d of up to 4 limbs, schoolbook bit-by-bit (runs once)
inline std::vector<uint64_t> mp_div(const std::vector<uint64_t>& a, const std::vector<uint64_t>& d, std::vector<uint64_t>* rem_out = nullptr) {
    std::vector<uint64_t> q(a.size(), 0), rem(d.size() + 1, 0);
    for (int bit = (int)a.size() * 64 - 1; bit >= 0; --bit) {
        uint64_t carry = (a[bit >> 6] >> (bit & 63)) & 1;             // rem = rem * 2 + bit
        for (size_t i = 0; i < rem.size(); ++i) { const uint64_t n = (rem[i] << 1) | carry; carry = rem[i] >> 63; rem[i] = n; }
        bool ge = true;
        for (int i = (int)rem.size() - 1; i >= 0; --i) {
            const uint64_t dv = (size_t)i < d.size() ? d[i] : 0;
            if (rem[i] != dv) { ge = rem[i] > dv; break; }
        }
        if (ge) {
            uint64_t br = 0;
            for (size_t i = 0; i < rem.size(); ++i) {
                const uint64_t dv = i < d.size() ? d[i] : 0;
                const u128 s = (u128)rem[i] - dv - br;
                rem[i] = (uint64_t)s;
                br = (uint64_t)(s >> 64) & 1;
            }
            q[bit >> 6] |= 1ull << (bit & 63);
        }
    }
    while (q.size() > 1 && q.back() == 0) q.pop_back();
    if (rem_out) *rem_out = rem;
    return q;
}
inline PairingConsts::PairingConsts() {
    const std::vector<uint64_t> p(QC::M, QC::M + 4), r(FrC::M, FrC::M + 4);
    const Fq2 xi{q_from_u64(9), q_one()};
    const std::vector<uint64_t> pm1 = mp_sub_small(p, 1);
    const std::vector<uint64_t> e3 = mp_div(pm1, {3}), e2 = mp_div(pm1, {2});
    frob_x = f2_pow(xi, e3.data(), (int)e3.size());
    frob_y = f2_pow(xi, e2.data(), (int)e2.size());
    std::vector<uint64_t> p12{1};
    for (int i = 0; i < 12; ++i) p12 = mp_mul(p12, p);
    final_exp = mp_div(mp_sub_small(p12, 1), r);
}
inline const PairingConsts& pairing_consts() { static const PairingConsts c; return c; }

// line through T and Q2 (tangent when they coincide) evaluated at P = (px, py); T <- T + Q2.
// In twisted coordinates the value is  -py + (lambda px) w + (y_T - lambda x_T) w^3  (module header).
inline Fq12 line_and_add(G2Affine* T, const G2Affine& Q2, const F4& px, const F4& py) {
    Fq2 lambda;
    Fq12 l;
    for (F4& v : l.c) v = q_zero();
    if (!f2_eq(T->x, Q2.x)) {
        lambda = f2_mul(f2_sub(Q2.y, T->y), f2_inv(f2_sub(Q2.x, T->x)));
    } else if (f2_eq(T->y, Q2.y)) {
        const Fq2 x2 = f2_mul(T->x, T->x);
        lambda = f2_mul(f2_add(f2_add(x2, x2), x2), f2_inv(f2_add(T->y, T->y)));
    } else {
        // vertical line x - x_T: value px - x_T w^2; the sum is the identity (does not occur for points of order r inside the loop)
        l.c[0] = px;
        f12_place(&l, f2_neg(T->x), 2);
        memset(T, 0, sizeof *T);
        return l;
    }
    l.c[0] = q_neg(py);
    f12_place(&l, f2_scale(lambda, px), 1);
    f12_place(&l, f2_sub(T->y, f2_mul(lambda, T->x)), 3);
    const Fq2 x3 = f2_sub(f2_sub(f2_mul(lambda, lambda), T->x), Q2.x);
    const Fq2 y3 = f2_sub(f2_mul(lambda, f2_sub(T->x, x3)), T->y);
    T->x = x3;
    T->y = y3;
    return l;
}
// Miller loop of the optimal ate pairing, no final exponentiation.  P, Q affine, neither the identity.
inline Fq12 miller_loop(const G1Affine& P, const G2Affine& Q) {
    // 6 t + 2 = 29793968203157093288 = 2^64 + 11347224129447541672: the top bit is the start T = Q, bits 63..0 drive the loop
    static const uint64_t ATE = 11347224129447541672ull;
    static const int TOP = 63;
    F4 px, py;
    memcpy(px.l, &P.x, 32);
    memcpy(py.l, &P.y, 32);
    G2Affine T = Q;
    Fq12 f = f12_one();
    for (int i = TOP; i >= 0; --i) {
        const Fq12 l = line_and_add(&T, T, px, py);
        f = f12_mul(f12_mul(f, f), l);
        if ((ATE >> i) & 1) f = f12_mul(f, line_and_add(&T, Q, px, py));
    }
    const PairingConsts& c = pairing_consts();
    const G2Affine Q1{f2_mul(f2_conj(Q.x), c.frob_x), f2_mul(f2_conj(Q.y), c.frob_y)};
    const G2Affine nQ2{f2_mul(f2_conj(Q1.x), c.frob_x), f2_neg(f2_mul(f2_conj(Q1.y), c.frob_y))};
    f = f12_mul(f, line_and_add(&T, Q1, px, py));
    f = f12_mul(f, line_and_add(&T, nQ2, px, py));
    return f;
}
inline Fq12 final_exponentiation(const Fq12& f) {
    const PairingConsts& c = pairing_consts();
    return f12_pow(f, c.final_exp.data(), (int)c.final_exp.size());
}
// prod_i e(P_i, Q_i) == 1, one shared final exponentiation (pairs with an identity contribute 1)
inline bool pairing_check(const G1Affine* P, const G2Affine* Q, size_t n) {
    Fq12 f = f12_one();
    for (size_t i = 0; i < n; ++i) {
        if (P[i].is_identity() || g2_is_identity(Q[i])) continue;
        f = f12_mul(f, miller_loop(P[i], Q[i]));
    }
    return f12_eq(final_exponentiation(f), f12_one());
}

// --------------------------------------------------------------------------------- G1 on the host
inline PXyzz g1_lift(const G1Affine& p) {
    PXyzz q;
    memset(&q, 0, sizeof q);
    if (p.is_identity()) return q;
    memcpy(&q.x, &p.x, 32);
    memcpy(&q.y, &p.y, 32);
    q.zz = q_one();
    q.zzz = q_one();
    return q;
}
// k * P for a canonical (non-Montgomery) 256-bit scalar, double-and-add from the top bit
inline PXyzz g1_mul_canon(const G1Affine& p, const F4& k) {
    const PXyzz base = g1_lift(p);
    PXyzz acc;
    memset(&acc, 0, sizeof acc);
    for (int i = 255; i >= 0; --i) {
        acc = pdbl(acc);
        if ((k.l[i >> 6] >> (i & 63)) & 1) acc = padd(acc, base);
    }
    return acc;
}
// y from x on y^2 = x^3 + 3 (p = 3 mod 4: y = (x^3 + 3)^((p+1)/4)); false when x is not on the curve
inline bool g1_y_from_x(const F4& x, F4* y) {
    const F4 rhs = q_add(q_mul(q_mul(x, x), x), q_from_u64(3));
    std::vector<uint64_t> e(QC::M, QC::M + 4);
    e[0] += 1;                                                    // p + 1 (no carry: p ends in ...47)
    for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 2) | (i + 1 < 4 ? e[i + 1] << 62 : 0);
    F4 r = q_one(), b = rhs;
    for (int i = 0; i < 256; ++i) { if ((e[i >> 6] >> (i & 63)) & 1) r = q_mul(r, b); b = q_mul(b, b); }
    *y = r;
    return memcmp(q_mul(r, r).l, rhs.l, 32) == 0;
}

}  // namespace host
}  // namespace zk
