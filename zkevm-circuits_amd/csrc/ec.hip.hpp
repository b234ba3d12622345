// BN254 G1 (y^2 = x^3 + 3 over Fq) group law for gfx950.
//
// Replaces (on device): halo2curves 0.1.0 src/bn256/curve.rs + src/derive/curve.rs
// (`new_curve_impl!`: G1Affine{x,y} with identity (0,0); G1 Jacobian {x,y,z})  [EXT, SURVEY 8a K1]
//
// Bucket / partial-sum state uses extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity: ZZ = 0): the mixed addition costs 8M + 2S with
// no inversion and the representation never needs a Z on its own.  Results leave the device as
// XYZZ and are normalised to affine (the only form the transcript sees) on the host.
#pragma once
#include "ff.hip.hpp"

namespace zk {

struct alignas(16) G1Affine {
    Fq x, y;
    __host__ __device__ bool is_identity() const { return x.is_zero() && y.is_zero(); }
};

struct alignas(16) G1Xyzz {
    Fq x, y, zz, zzz;
    __host__ __device__ static G1Xyzz identity() { return G1Xyzz{Fq::zero(), Fq::zero(), Fq::zero(), Fq::zero()}; }
    __host__ __device__ bool is_identity() const { return zz.is_zero(); }
};

// Jacobian image used at the ABI (halo2curves `G1` layout: x, y, z)
struct alignas(16) G1Jac {
    Fq x, y, z;
};

__host__ __device__ __forceinline__ G1Xyzz from_affine(const G1Affine& p) {
    if (p.is_identity()) return G1Xyzz::identity();
    return G1Xyzz{p.x, p.y, Fq::one(), Fq::one()};
}

// 2 * (affine P)   (mdbl-2008-s-1)
__host__ __device__ __forceinline__ G1Xyzz dbl_affine(const G1Affine& p) {
    if (p.is_identity() || p.y.is_zero()) return G1Xyzz::identity();
    Fq u = dbl(p.y), v = sqr(u), w = u * v, s = p.x * v;
    Fq x2 = sqr(p.x);
    Fq m = dbl(x2) + x2;
    G1Xyzz r;
    r.x = sqr(m) - dbl(s);
    r.y = m * (s - r.x) - w * p.y;
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2 * P   (dbl-2008-s-1, a = 0)
__host__ __device__ __forceinline__ G1Xyzz dbl(const G1Xyzz& p) {
    if (p.is_identity() || p.y.is_zero()) return G1Xyzz::identity();
    Fq u = dbl(p.y), v = sqr(u), w = u * v, s = p.x * v;
    Fq x2 = sqr(p.x);
    Fq m = dbl(x2) + x2;
    G1Xyzz r;
    r.x = sqr(m) - dbl(s);
    r.y = m * (s - r.x) - w * p.y;
    r.zz = v * p.zz;
    r.zzz = w * p.zzz;
    return r;
}

// P + (affine Q)   (madd-2008-s), all exceptional cases handled
__host__ __device__ __forceinline__ G1Xyzz madd(const G1Xyzz& p, const G1Affine& q) {
    if (q.is_identity()) return p;
    if (p.is_identity()) return G1Xyzz{q.x, q.y, Fq::one(), Fq::one()};
    Fq u2 = q.x * p.zz, s2 = q.y * p.zzz;
    Fq pp_ = u2 - p.x, r_ = s2 - p.y;
    if (pp_.is_zero()) {
        if (r_.is_zero()) return dbl_affine(q);
        return G1Xyzz::identity();
    }
    Fq pp = sqr(pp_), ppp = pp_ * pp, qq = p.x * pp;
    G1Xyzz r;
    r.x = sqr(r_) - ppp - dbl(qq);
    r.y = r_ * (qq - r.x) - p.y * ppp;
    r.zz = p.zz * pp;
    r.zzz = p.zzz * ppp;
    return r;
}

// P + Q   (add-2008-s), all exceptional cases handled
__host__ __device__ __forceinline__ G1Xyzz add(const G1Xyzz& p, const G1Xyzz& q) {
    if (q.is_identity()) return p;
    if (p.is_identity()) return q;
    Fq u1 = p.x * q.zz, u2 = q.x * p.zz, s1 = p.y * q.zzz, s2 = q.y * p.zzz;
    Fq pp_ = u2 - u1, r_ = s2 - s1;
    if (pp_.is_zero()) {
        if (r_.is_zero()) return dbl(p);
        return G1Xyzz::identity();
    }
    Fq pp = sqr(pp_), ppp = pp_ * pp, qq = u1 * pp;
    G1Xyzz r;
    r.x = sqr(r_) - ppp - dbl(qq);
    r.y = r_ * (qq - r.x) - s1 * ppp;
    r.zz = p.zz * q.zz * pp;
    r.zzz = p.zzz * q.zzz * ppp;
    return r;
}

__host__ __device__ __forceinline__ G1Affine neg(const G1Affine& p) { return G1Affine{p.x, neg(p.y)}; }
__host__ __device__ __forceinline__ G1Xyzz neg(const G1Xyzz& p) { return G1Xyzz{p.x, neg(p.y), p.zz, p.zzz}; }

// XYZZ -> affine (one field inversion): x = X/ZZ, y = Y/ZZZ
__host__ __device__ inline G1Affine to_affine(const G1Xyzz& p) {
    if (p.is_identity()) return G1Affine{Fq::zero(), Fq::zero()};
    // 1/ZZZ gives both: 1/ZZ = (1/ZZZ)^2 * ZZ^2 ... cheaper: inv(zz*zzz) then split
    Fq t = inv(p.zz * p.zzz);
    Fq izz = t * p.zzz, izzz = t * p.zz;
    return G1Affine{p.x * izz, p.y * izzz};
}
// XYZZ -> Jacobian with Z = ZZZ/ZZ:  X_j = X * Z^2 / ZZ = X * (ZZZ/ZZ)^2 / ZZ ... use affine route
__host__ __device__ inline G1Jac to_jacobian(const G1Xyzz& p) {
    if (p.is_identity()) return G1Jac{Fq::zero(), Fq::one(), Fq::zero()};
    G1Affine a = to_affine(p);
    return G1Jac{a.x, a.y, Fq::one()};
}

__device__ __forceinline__ G1Affine ldg(const G1Affine* p) {
    const Fq* f = reinterpret_cast<const Fq*>(p);
    return G1Affine{ldg(f), ldg(f + 1)};
}
__device__ __forceinline__ G1Xyzz ldg(const G1Xyzz* p) {
    const Fq* f = reinterpret_cast<const Fq*>(p);
    return G1Xyzz{ldg(f), ldg(f + 1), ldg(f + 2), ldg(f + 3)};
}
__device__ __forceinline__ void stg(G1Xyzz* p, const G1Xyzz& v) {
    Fq* f = reinterpret_cast<Fq*>(p);
    stg(f, v.x); stg(f + 1, v.y); stg(f + 2, v.zz); stg(f + 3, v.zzz);
}
__device__ __forceinline__ void stg(G1Affine* p, const G1Affine& v) {
    Fq* f = reinterpret_cast<Fq*>(p);
    stg(f, v.x); stg(f + 1, v.y);
}

}  // namespace zk
