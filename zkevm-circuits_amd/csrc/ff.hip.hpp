// BN254 prime-field arithmetic for gfx950 (CDNA4): 8 x u32 little-endian limbs, Montgomery form
// (R = 2^256).  The in-memory image of an element is byte-identical to halo2curves'
// 4 x u64 Montgomery limbs (SURVEY.md 8b "Data layout at the boundary"), so columns, SRS points
// and proving-key polynomials are consumed without conversion.
//
// Replaces (on device): halo2curves 0.1.0 src/bn256/{fr,fq}.rs field_arithmetic!/mont  [EXT]
// Reference call sites that reach it: SURVEY.md 8a K1-K12.
//
// Design notes (gfx950):
//   * v_mad_u64_u32 (32x32+64 -> 64) is the multiplier primitive; everything is written so the
//     compiler emits it with the accumulator as the 64-bit addend.
//   * both moduli are < 2^254, so a + b never carries out of limb 7 and the CIOS running sum
//     never needs a 10th limb ("no-carry" CIOS).
//   * all public operations take and return canonical values in [0, m).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zk {

struct FrP {
    static constexpr uint32_t INV = 0xefffffffu;
    __host__ __device__ static constexpr uint32_t M(int i) {
        constexpr uint32_t m[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return m[i];
    }
    __host__ __device__ static constexpr uint32_t ONE(int i) {   // R mod m
        constexpr uint32_t r[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return r[i];
    }
    __host__ __device__ static constexpr uint32_t R2(int i) {    // R^2 mod m
        constexpr uint32_t r[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return r[i];
    }
};
struct FqP {
    static constexpr uint32_t INV = 0xe4866389u;
    __host__ __device__ static constexpr uint32_t M(int i) {
        constexpr uint32_t m[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return m[i];
    }
    __host__ __device__ static constexpr uint32_t ONE(int i) {
        constexpr uint32_t r[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return r[i];
    }
    __host__ __device__ static constexpr uint32_t R2(int i) {
        constexpr uint32_t r[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return r[i];
    }
};

template <class P>
struct alignas(16) Fp {
    uint32_t l[8];

    __host__ __device__ static constexpr Fp zero() { return Fp{{0, 0, 0, 0, 0, 0, 0, 0}}; }
    __host__ __device__ static constexpr Fp one() {
        return Fp{{P::ONE(0), P::ONE(1), P::ONE(2), P::ONE(3), P::ONE(4), P::ONE(5), P::ONE(6), P::ONE(7)}};
    }
    __host__ __device__ static constexpr Fp r2() {
        return Fp{{P::R2(0), P::R2(1), P::R2(2), P::R2(3), P::R2(4), P::R2(5), P::R2(6), P::R2(7)}};
    }
    __host__ __device__ bool is_zero() const { return (l[0] | l[1] | l[2] | l[3] | l[4] | l[5] | l[6] | l[7]) == 0; }
    __host__ __device__ bool operator==(const Fp& o) const {
        uint32_t d = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) d |= l[i] ^ o.l[i];
        return d == 0;
    }
    __host__ __device__ bool operator!=(const Fp& o) const { return !(*this == o); }
};

// ---- limb helpers ---------------------------------------------------------------------------
// r = a + b over 8 limbs, returns carry out.  __builtin_addc lowers to one v_add_co_u32 +
// seven v_addc_co_u32 on gfx950 (checked in the ISA); a u64-accumulator formulation instead
// produces v_lshl_add_u64 + v_mov pairs (3x the instructions).
__host__ __device__ __forceinline__ uint32_t add8(uint32_t (&r)[8], const uint32_t (&a)[8], const uint32_t (&b)[8]) {
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { unsigned co; r[i] = __builtin_addc(a[i], b[i], c, &co); c = co; }
    return c;
}
// r = a - b over 8 limbs, returns borrow (0/1)
__host__ __device__ __forceinline__ uint32_t sub8(uint32_t (&r)[8], const uint32_t (&a)[8], const uint32_t (&b)[8]) {
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { unsigned co; r[i] = __builtin_subc(a[i], b[i], c, &co); c = co; }
    return c;
}
template <class P>
__host__ __device__ __forceinline__ void load_mod(uint32_t (&m)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = P::M(i);
}
// if (x >= m) x -= m          (x < 2m)
template <class P>
__host__ __device__ __forceinline__ void cond_sub(uint32_t (&x)[8]) {
    uint32_t m[8], t[8];
    load_mod<P>(m);
    uint32_t borrow = sub8(t, x, m);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = borrow ? x[i] : t[i];
}

// x^-1 mod m for a plain integer 0 < x < m (NOT a Montgomery residue): binary extended Euclid, invariants a x = u, b x = v (mod m).
// About 1.4 x 254 halvings and 0.7 x 254 subtractions of 8-limb integers -- a quarter of the instructions of the 381-product
// Fermat ladder, which is what counts where ONE lane computes an inverse and its wave waits (batch inversion, vec.hip).
// Data-dependent trip counts: the running time depends on the value.  The one caller inverts products of a whole wave's worth of
// column values inside a proving kernel, where timing is not an interface; do not use it on a lone secret.  x = 0 returns 0.
template <class P>
__host__ __device__ inline void inv_xgcd(uint32_t (&out)[8], const uint32_t (&x)[8]) {
    uint32_t u[8], v[8], a[8], b[8], m[8], t[8];
    load_mod<P>(m);
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { u[i] = x[i]; v[i] = m[i]; a[i] = i == 0 ? 1u : 0u; b[i] = 0u; any |= x[i]; }
    if (!any) {
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] = 0u;
        return;
    }
    auto is_one = [](const uint32_t (&w)[8]) { return w[0] == 1u && (w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) == 0u; };
    auto halve = [](uint32_t (&w)[8]) {
#pragma unroll
        for (int i = 0; i < 7; ++i) w[i] = (w[i] >> 1) | (w[i + 1] << 31);
        w[7] >>= 1;
    };
    for (int guard = 0; guard < 1024 && !is_one(u) && !is_one(v); ++guard) {
        while (!(u[0] & 1u)) { halve(u); if (a[0] & 1u) add8(a, a, m); halve(a); }          // a < m < 2^254: a + m fits
        while (!(v[0] & 1u)) { halve(v); if (b[0] & 1u) add8(b, b, m); halve(b); }
        if (sub8(t, u, v) == 0u) {                 // u >= v
#pragma unroll
            for (int i = 0; i < 8; ++i) u[i] = t[i];
            if (sub8(a, a, b)) add8(a, a, m);
        } else {
            sub8(v, v, u);
            if (sub8(b, b, a)) add8(b, b, m);
        }
    }
    const bool from_u = is_one(u);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = from_u ? a[i] : b[i];
}

template <class P>
__host__ __device__ __forceinline__ Fp<P> operator+(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    add8(r.l, a.l, b.l);       // no carry out: a,b < m < 2^254
    cond_sub<P>(r.l);
    return r;
}
template <class P>
__host__ __device__ __forceinline__ Fp<P> operator-(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint32_t t[8], m[8];
    load_mod<P>(m);
    uint32_t borrow = sub8(r.l, a.l, b.l);
    add8(t, r.l, m);
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = borrow ? t[i] : r.l[i];
    return r;
}
template <class P>
__host__ __device__ __forceinline__ Fp<P> neg(const Fp<P>& a) {
    Fp<P> r;
    uint32_t m[8];
    load_mod<P>(m);
    sub8(r.l, m, a.l);
    const bool z = a.is_zero();
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = z ? 0u : r.l[i];
    return r;
}
template <class P>
__host__ __device__ __forceinline__ Fp<P> dbl(const Fp<P>& a) { return a + a; }

// ---- Montgomery product: CIOS, 32-bit limbs, fused multiply/reduce rows ("no-carry" form) -----
// Per row i:  t += a * b[i];  m = t[0] * INV;  t = (t + m * M) >> 32.
// Invariant (m < 2^254, a,b < m): t < 2m after every row, so t fits 8 limbs + the row carry.
template <class P>
__host__ __device__ __forceinline__ Fp<P> operator*(const Fp<P>& a, const Fp<P>& b) {
    uint32_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t bi = b.l[i];
        // first column decides m
        uint64_t c1 = (uint64_t)a.l[0] * bi + t[0];
        const uint32_t m = (uint32_t)c1 * P::INV;
        uint64_t c2 = (uint64_t)m * P::M(0) + (uint32_t)c1;
        c1 >>= 32;
        c2 >>= 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            c1 += (uint64_t)a.l[j] * bi + t[j];
            c2 += (uint64_t)m * P::M(j) + (uint32_t)c1;
            t[j - 1] = (uint32_t)c2;
            c1 >>= 32;
            c2 >>= 32;
        }
        t[7] = (uint32_t)(c1 + c2);
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = t[i];
    cond_sub<P>(r.l);
    return r;
}
template <class P>
__host__ __device__ __forceinline__ Fp<P> sqr(const Fp<P>& a) { return a * a; }

// Montgomery form <-> canonical integer
template <class P>
__host__ __device__ __forceinline__ Fp<P> to_mont(const Fp<P>& a) { return a * Fp<P>::r2(); }
template <class P>
__host__ __device__ __forceinline__ Fp<P> from_mont(const Fp<P>& a) {
    Fp<P> one = Fp<P>::zero();
    one.l[0] = 1;
    return a * one;
}

template <class P>
__host__ __device__ inline Fp<P> pow_u64(Fp<P> base, uint64_t e) {
    Fp<P> r = Fp<P>::one();
    while (e) {
        if (e & 1) r = r * base;
        base = sqr(base);
        e >>= 1;
    }
    return r;
}
// a^(m-2); inv(0) = 0
template <class P>
__host__ __device__ inline Fp<P> inv(const Fp<P>& a) {
    uint32_t e[8];
    load_mod<P>(e);
    e[0] -= 2;   // both moduli have low limb >= 2
    Fp<P> r = Fp<P>::one(), b = a;
    for (int i = 0; i < 254; ++i) {
        if ((e[i >> 5] >> (i & 31)) & 1) r = r * b;
        b = sqr(b);
    }
    return r;
}

using Fr = Fp<FrP>;
using Fq = Fp<FqP>;

// ---- global-memory access: one element = two 16-byte transactions ---------------------------
template <class F>
__device__ __forceinline__ F ldg(const F* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    F r;
    r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
    r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
    return r;
}
template <class F>
__device__ __forceinline__ void stg(F* p, const F& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

}  // namespace zk
