// BN254 G1 group law on 9 x 29-bit limbs (ff29.hip.hpp), used by the MSM inner loops.
// Same formulas as ec.hip.hpp (XYZZ: madd-2008-s / add-2008-s / dbl-2008-s-1, a = 0); coordinates are
// held in R' = 2^261 Montgomery form, lazily reduced:
//
//   stored invariant:  every limb 0..7 < 2^29 ("normalised"); x, y < 8p;  zz, zzz < 2p
//   identity:          zz has all limbs zero (never produced from non-identity inputs)
//   affine input:      canonical (< p), normalised, R' form; identity = (0, 0)
//
// Bound bookkeeping uses mul29's guarantee  out < a*b/2^261 + p  (p/2^261 < 0.006), so any
// product of values below ~16p comes out below 2.6p, and sub29k<K> needs its subtrahend < K*p.
#pragma once
#include "ec.hip.hpp"
#include "ff29.hip.hpp"

// 1: y3 of every addition / doubling as one two-product Montgomery pass (mul2add29) and the quick k*p filter in front of
// is_zero_mod_p29; 0: the round-1 formulation (two reduced products and a subtraction), kept for A/B builds.
#ifndef ZK_EC_FUSED
#define ZK_EC_FUSED 1
#endif

namespace zk {

struct G1Affine29 {
    Fq29 x, y;
};
struct alignas(16) G1Xyzz29 {
    Fq29 x, y, zz, zzz;    // 36 x u32 = 144 bytes
};

__host__ __device__ __forceinline__ bool all_zero29(const Fq29& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) o |= a.l[i];
    return o == 0;
}
__host__ __device__ __forceinline__ Fq29 zero29() {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = 0;
    return r;
}
// R' image of 1: 2^261 mod p
__host__ __device__ __forceinline__ Fq29 one29() {
    // 2^261 mod p = 32 * (2^256 mod p) mod p, computed once at compile time would need big-int
    // constexpr; instead unpack the 8x32 constant produced by five doublings of Fq::one().
    Fq o = Fq::one();
#pragma unroll
    for (int i = 0; i < 5; ++i) o = dbl(o);
    return unpack29<Fq29P>(o);
}
__host__ __device__ __forceinline__ G1Xyzz29 identity29() { return G1Xyzz29{zero29(), zero29(), zero29(), zero29()}; }
__host__ __device__ __forceinline__ bool is_identity29(const G1Xyzz29& p) { return all_zero29(p.zz); }
__host__ __device__ __forceinline__ bool is_identity29(const G1Affine29& p) { return all_zero29(p.x) && all_zero29(p.y); }

// v normalised, 0 <= v < (MAXK + 1) * p :  v == 0 (mod p) ?
#if ZK_EC_FUSED
// v = k*p forces v.l[0] = k * p.l[0] (mod 2^29), so k = v.l[0] / p.l[0] (mod 2^29) = -(v.l[0] * INV) is the only candidate:
// its multiple of p is rebuilt limb by limb and compared (9 multiply-adds, no branch, no table of multiples).
template <int MAXK>
__host__ __device__ __forceinline__ bool is_zero_mod_p29(const Fq29& v) {
    const uint32_t k = (0u - v.l[0] * Fq29P::INV) & MASK29;
    uint32_t diff = 0;
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        acc += (uint64_t)k * Fq29P::M(i);
        diff |= v.l[i] ^ (i < 8 ? (uint32_t)acc & MASK29 : (uint32_t)acc);
        acc >>= 29;
    }
    return diff == 0 && k <= (uint32_t)MAXK;
}
#else
template <int MAXK>
__host__ __device__ __forceinline__ bool is_zero_mod_p29(const Fq29& v) {
    bool any = all_zero29(v);
#pragma unroll
    for (int k = 1; k <= MAXK; ++k) {
        // normalised limbs of k*p = balanced limbs with the borrow undone
        uint32_t diff = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            uint32_t c = kp_balanced<Fq29P>(k, i);
            if (i < 8) c -= 1u << 29;
            if (i > 0) c += 1u;
            diff |= v.l[i] ^ c;
        }
        any = any || diff == 0;
    }
    return any;
}
#endif

__host__ __device__ __forceinline__ Fq29 add_n(const Fq29& a, const Fq29& b) { Fq29 r = add29(a, b); normalize29(r); return r; }
template <int K>
__host__ __device__ __forceinline__ Fq29 sub_n(const Fq29& a, const Fq29& b) { Fq29 r = sub29k<K>(a, b); normalize29(r); return r; }
// a - b + K p for a subtrahend that is the plain limb-wise sum of S normalised values (not carried): S borrows per limb
template <int K, int S>
__host__ __device__ __forceinline__ Fq29 sub_nw(const Fq29& a, const Fq29& b) { Fq29 r = sub29kw<K, S>(a, b); normalize29(r); return r; }

// -y for a canonical affine y (y != 0 for points on the curve; y = 0 maps to p == 0 mod p)
__host__ __device__ __forceinline__ Fq29 neg_canon29(const Fq29& y) { return sub_n<1>(zero29(), y); }

// 2 * (affine Q)  (mdbl-2008-s-1)
__host__ __device__ __forceinline__ G1Xyzz29 dbl_affine29(const G1Affine29& q) {
    Fq29 u = add_n(q.y, q.y);                // < 2p
    Fq29 v = sqr29(u), w = mul29(u, v), s = mul29(q.x, v);
    Fq29 x2 = sqr29(q.x);
    Fq29 m = add_n(add29(x2, x2), x2);       // 3 x^2 < 3.1p
    G1Xyzz29 r;
#if ZK_EC_FUSED
    r.x = sub_nw<3, 2>(sqr29(m), add29(s, s));             // < 1.1p + 3p; the sum 2s is subtracted as it is, not carried first
#else
    r.x = sub_n<3>(sqr29(m), add_n(s, s));                 // < 1.1p + 3p
#endif
#if ZK_EC_FUSED
    r.y = mul2add29(m, sub_n<5>(s, r.x), q.y, neg29k<2>(w));    // m (s - x3) + y (2p - w) < (3.1 * 6.1 + 2) p^2: below 1.2p
#else
    r.y = sub_n<2>(mul29(m, sub_n<5>(s, r.x)), mul29(w, q.y)); // s - x3 + 5p < 6.1p
#endif
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2 * P  (dbl-2008-s-1)
__host__ __device__ __forceinline__ G1Xyzz29 dbl29pt(const G1Xyzz29& p) {
    if (is_identity29(p)) return p;
    Fq29 u = add_n(p.y, p.y);                // < 16p
    Fq29 v = sqr29(u), w = mul29(u, v), s = mul29(p.x, v);   // v < 2.6p, w < 1.3p, s < 1.2p
    Fq29 x2 = sqr29(p.x);               // < 1.4p
    Fq29 m = add_n(add29(x2, x2), x2);       // < 4.2p
    G1Xyzz29 r;
#if ZK_EC_FUSED
    r.x = sub_nw<3, 2>(sqr29(m), add29(s, s));               // < 1.2p + 3p
#else
    r.x = sub_n<3>(sqr29(m), add_n(s, s));                   // < 1.2p + 3p
#endif
#if ZK_EC_FUSED
    r.y = mul2add29(m, sub_n<5>(s, r.x), p.y, neg29k<2>(w));    // (4.2 * 6.2 + 8 * 2) p^2 / 2^261 + p: below 1.3p
#else
    r.y = sub_n<2>(mul29(m, sub_n<5>(s, r.x)), mul29(w, p.y));  // (s - x3 + 5p) < 6.2p
#endif
    r.zz = mul29(v, p.zz);
    r.zzz = mul29(w, p.zzz);
    return r;
}

// P + (affine Q)  (madd-2008-s), all exceptional cases handled
__host__ __device__ __forceinline__ G1Xyzz29 madd29(const G1Xyzz29& p, const G1Affine29& q) {
    if (is_identity29(q)) return p;
    if (is_identity29(p)) { const Fq29 one = one29(); return G1Xyzz29{q.x, q.y, one, one}; }
    Fq29 u2 = mul29(q.x, p.zz), s2 = mul29(q.y, p.zzz);    // < 1.1p
    Fq29 pd = sub_n<8>(u2, p.x), rd = sub_n<8>(s2, p.y);   // < 9.1p
    if (is_zero_mod_p29<9>(pd)) {
        if (is_zero_mod_p29<9>(rd)) return dbl_affine29(q);
        return identity29();
    }
    Fq29 pp = sqr29(pd), ppp = mul29(pd, pp), qq = mul29(p.x, pp);   // < 1.5p, 1.1p, 1.1p
    G1Xyzz29 r;
#if ZK_EC_FUSED
    r.x = sub_nw<4, 3>(sqr29(rd), add29(add29(ppp, qq), qq));         // < 1.5p + 4p
#else
    r.x = sub_n<4>(sqr29(rd), add_n(add29(ppp, qq), qq));            // < 1.5p + 4p
#endif
#if ZK_EC_FUSED
    r.y = mul2add29(rd, sub_n<6>(qq, r.x), p.y, neg29k<2>(ppp));      // (9.1 * 7.2 + 8 * 2) p^2 / 2^261 + p: below 1.5p
#else
    r.y = sub_n<2>(mul29(rd, sub_n<6>(qq, r.x)), mul29(p.y, ppp));       // < 1.4p + 2p
#endif
    r.zz = mul29(p.zz, pp);
    r.zzz = mul29(p.zzz, ppp);
    return r;
}

// P + Q  (add-2008-s), all exceptional cases handled
__host__ __device__ __forceinline__ G1Xyzz29 add29pt(const G1Xyzz29& p, const G1Xyzz29& q) {
    if (is_identity29(q)) return p;
    if (is_identity29(p)) return q;
    Fq29 u1 = mul29(p.x, q.zz), u2 = mul29(q.x, p.zz), s1 = mul29(p.y, q.zzz), s2 = mul29(q.y, p.zzz);   // < 1.1p
    Fq29 pd = sub_n<2>(u2, u1), rd = sub_n<2>(s2, s1);     // < 3.1p
    if (is_zero_mod_p29<3>(pd)) {
        if (is_zero_mod_p29<3>(rd)) return dbl29pt(p);
        return identity29();
    }
    Fq29 pp = sqr29(pd), ppp = mul29(pd, pp), qq = mul29(u1, pp);
    G1Xyzz29 r;
#if ZK_EC_FUSED
    r.x = sub_nw<4, 3>(sqr29(rd), add29(add29(ppp, qq), qq));
#else
    r.x = sub_n<4>(sqr29(rd), add_n(add29(ppp, qq), qq));
#endif
#if ZK_EC_FUSED
    r.y = mul2add29(rd, sub_n<6>(qq, r.x), s1, neg29k<2>(ppp));
#else
    r.y = sub_n<2>(mul29(rd, sub_n<6>(qq, r.x)), mul29(s1, ppp));
#endif
    r.zz = mul29(mul29(p.zz, q.zz), pp);
    r.zzz = mul29(mul29(p.zzz, q.zzz), ppp);
    return r;
}

// ---- conversions ------------------------------------------------------------------------------
// canonical 8x32 affine in R' form -> limbs
__device__ __forceinline__ G1Affine29 load_affine29(const G1Affine* p) {
    const Fq* f = reinterpret_cast<const Fq*>(p);
    return G1Affine29{unpack29<Fq29P>(ldg(f)), unpack29<Fq29P>(ldg(f + 1))};
}
__device__ __forceinline__ G1Xyzz29 ldg29(const G1Xyzz29* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint32_t w[36];
#pragma unroll
    for (int i = 0; i < 9; ++i) { uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
    G1Xyzz29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) { r.x.l[i] = w[i]; r.y.l[i] = w[9 + i]; r.zz.l[i] = w[18 + i]; r.zzz.l[i] = w[27 + i]; }
    return r;
}
__device__ __forceinline__ void stg29(G1Xyzz29* p, const G1Xyzz29& v) {
    uint32_t w[36];
#pragma unroll
    for (int i = 0; i < 9; ++i) { w[i] = v.x.l[i]; w[9 + i] = v.y.l[i]; w[18 + i] = v.zz.l[i]; w[27 + i] = v.zzz.l[i]; }
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < 9; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
// a^(p-2) in the R' Montgomery domain (a normalised, < 2^258); result normalised, < 2p
__device__ inline Fq29 inv29(const Fq29& a) {
    Fq29 r = one29(), b = a;
#pragma unroll 1
    for (int i = 0; i < 254; ++i) {
        // exponent p - 2: bits of the modulus with the low limb reduced by 2 (p = ...fd47, so no borrow)
        const uint32_t limb = Fq29P::P32::M(i >> 5) - (i < 32 ? 2u : 0u);
        if ((limb >> (i & 31)) & 1) r = mul29(r, b);
        b = sqr29(b);
    }
    return r;
}
// XYZZ (R', lazy) -> canonical affine in R' form, 8 x 32 limbs (the MSM kernels' base format)
__device__ inline G1Affine to_affine_rp(const G1Xyzz29& p) {
    if (is_identity29(p)) return G1Affine{Fq::zero(), Fq::zero()};
    const Fq29 t = inv29(mul29(p.zz, p.zzz));
    const Fq29 izz = mul29(t, p.zzz), izzz = mul29(t, p.zz);
    return G1Affine{pack29_lt2p(mul29(p.x, izz)), pack29_lt2p(mul29(p.y, izzz))};
}

// lazily reduced R' coordinates -> canonical R = 2^256 Montgomery XYZZ (what the host tail reads):
// x_R = x' * 2^256 / 2^261  (one product by the plain integer 2^256 mod p)
__device__ __forceinline__ G1Xyzz to_std_xyzz(const G1Xyzz29& p) {
    if (is_identity29(p)) return G1Xyzz::identity();
    const Fq29 c = unpack29<Fq29P>(Fq::one());   // limbs of the integer 2^256 mod p
    G1Xyzz r;
    r.x = pack29_lt2p(mul29(p.x, c));
    r.y = pack29_lt2p(mul29(p.y, c));
    r.zz = pack29_lt2p(mul29(p.zz, c));
    r.zzz = pack29_lt2p(mul29(p.zzz, c));
    return r;
}

}  // namespace zk
