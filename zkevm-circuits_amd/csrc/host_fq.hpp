// Host-side (x86-64) BN254 arithmetic for the short sequential tails that do not belong on a GPU
// lane: the Horner combination of the per-window MSM sums and the final XYZZ -> affine
// normalisation (one field inversion).  4 x u64 Montgomery limbs with unsigned __int128 --
// byte-compatible with the device Fp<> (8 x u32 LE).  Product code (not the test oracle).
#pragma once
#include <cstdint>
#include <cstring>

#include "ec.hip.hpp"

namespace zk {
namespace host {

typedef unsigned __int128 u128;
struct F4 { uint64_t l[4]; };

struct FqC {
    static constexpr uint64_t M[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t INV = 0x87d20782e4866389ULL;
    static constexpr uint64_t ONE[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
};
struct FrC {
    static constexpr uint64_t M[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t INV = 0xc2e1f593efffffffULL;
    static constexpr uint64_t ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
};

template <class C> inline bool geq_mod(const uint64_t* a) {
    for (int i = 3; i >= 0; --i) { if (a[i] > C::M[i]) return true; if (a[i] < C::M[i]) return false; }
    return true;
}
template <class C> inline void sub_mod(uint64_t* a) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - C::M[i] - (uint64_t)br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
template <class C> inline F4 fadd(const F4& a, const F4& b) {
    F4 r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (geq_mod<C>(r.l)) sub_mod<C>(r.l);
    return r;
}
template <class C> inline F4 fsub(const F4& a, const F4& b) {
    F4 r; u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a.l[i] - b.l[i] - (uint64_t)br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)r.l[i] + C::M[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
template <class C> inline F4 fmul(const F4& a, const F4& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * C::INV;
        c = (u128)m * C::M[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * C::M[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    F4 r; memcpy(r.l, t, 32);
    if (t[4] || geq_mod<C>(r.l)) sub_mod<C>(r.l);
    return r;
}
template <class C> inline bool fzero(const F4& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
template <class C> inline F4 fone() { F4 r; memcpy(r.l, C::ONE, 32); return r; }
template <class C> inline F4 finv(const F4& a) {     // a^(m-2)
    uint64_t e[4] = {C::M[0] - 2, C::M[1], C::M[2], C::M[3]};
    F4 r = fone<C>(), b = a;
    for (int i = 0; i < 256; ++i) { if ((e[i >> 6] >> (i & 63)) & 1) r = fmul<C>(r, b); b = fmul<C>(b, b); }
    return r;
}

struct PXyzz { F4 x, y, zz, zzz; };
static_assert(sizeof(PXyzz) == sizeof(G1Xyzz), "layout");

inline bool pid(const PXyzz& p) { return fzero<FqC>(p.zz); }
inline PXyzz pdbl(const PXyzz& p) {
    if (pid(p)) return p;
    typedef FqC C;
    F4 u = fadd<C>(p.y, p.y), v = fmul<C>(u, u), w = fmul<C>(u, v), s = fmul<C>(p.x, v);
    F4 x2 = fmul<C>(p.x, p.x), m = fadd<C>(fadd<C>(x2, x2), x2);
    PXyzz r;
    r.x = fsub<C>(fmul<C>(m, m), fadd<C>(s, s));
    r.y = fsub<C>(fmul<C>(m, fsub<C>(s, r.x)), fmul<C>(w, p.y));
    r.zz = fmul<C>(v, p.zz);
    r.zzz = fmul<C>(w, p.zzz);
    return r;
}
inline PXyzz padd(const PXyzz& p, const PXyzz& q) {
    if (pid(q)) return p;
    if (pid(p)) return q;
    typedef FqC C;
    F4 u1 = fmul<C>(p.x, q.zz), u2 = fmul<C>(q.x, p.zz), s1 = fmul<C>(p.y, q.zzz), s2 = fmul<C>(q.y, p.zzz);
    F4 P = fsub<C>(u2, u1), R = fsub<C>(s2, s1);
    if (fzero<C>(P)) {
        if (fzero<C>(R)) return pdbl(p);
        PXyzz id; memset(&id, 0, sizeof id); return id;
    }
    F4 pp = fmul<C>(P, P), ppp = fmul<C>(P, pp), qq = fmul<C>(u1, pp);
    PXyzz r;
    r.x = fsub<C>(fsub<C>(fmul<C>(R, R), ppp), fadd<C>(qq, qq));
    r.y = fsub<C>(fmul<C>(R, fsub<C>(qq, r.x)), fmul<C>(s1, ppp));
    r.zz = fmul<C>(fmul<C>(p.zz, q.zz), pp);
    r.zzz = fmul<C>(fmul<C>(p.zzz, q.zzz), ppp);
    return r;
}
inline void pto_affine(const PXyzz& p, G1Affine* out) {
    if (pid(p)) { memset(out, 0, sizeof(G1Affine)); return; }
    typedef FqC C;
    F4 t = finv<C>(fmul<C>(p.zz, p.zzz));
    F4 izz = fmul<C>(t, p.zzz), izzz = fmul<C>(t, p.zz);
    F4 x = fmul<C>(p.x, izz), y = fmul<C>(p.y, izzz);
    memcpy(&out->x, x.l, 32);
    memcpy(&out->y, y.l, 32);
}

// result = sum_w 2^(c*w) * S_w  (Horner from the top window), normalised to affine
inline void msm_tail(const G1Xyzz* wsum, int W, int c, G1Affine* out) {
    const PXyzz* s = reinterpret_cast<const PXyzz*>(wsum);
    PXyzz acc = s[W - 1];
    for (int w = W - 2; w >= 0; --w) {
        for (int i = 0; i < c; ++i) acc = pdbl(acc);
        acc = padd(acc, s[w]);
    }
    pto_affine(acc, out);
}

// sum of two XYZZ points (canonical R form) as an affine point: the main part of a column's MSM and the part of its blinded tail
inline void msm_tail2(const G1Xyzz* a, const G1Xyzz* b, G1Affine* out) {
    PXyzz acc = padd(*reinterpret_cast<const PXyzz*>(a), *reinterpret_cast<const PXyzz*>(b));
    pto_affine(acc, out);
}

}  // namespace host
}  // namespace zk
