// Unsaturated-limb BN254 field arithmetic for the ALU-bound inner loops (NTT butterflies, EC
// additions): 9 limbs x 29 bits, Montgomery radix R' = 2^261.
//
// Why (measured on MI355X, tools/ubench.hip): v_mad_u64_u32 issues at ~30 G lane-op/s x 1000 --
// the same rate as v_add_co/v_addc (carry-producing adds are half rate) -- so with saturated
// 32-bit limbs every 32x32 product needs a carry fix-up that costs as much as the multiply.
// With 29-bit limbs a 64-bit column accumulator absorbs all 18 products of a column without any
// carry handling: 162 pure v_mad_u64_u32 per Montgomery product and plain (carry-less) v_add_u32
// for field additions.
//
// Domain conventions:
//   * limbs are "normalised" when each is < 2^29 (top limb may carry the excess);
//   * values are kept lazily reduced (0 <= v < a few p); only pack() produces the canonical
//     representative;
//   * mul29(a, b) = a*b*2^-261 mod p, result normalised and < 2p provided a*b < 2^261 * p
//     (e.g. a < 2^258, b < 2^256).  Limb bounds: a[i] < 2^30, b[i] < 2^29 + small is safe
//     (column sum < 9*2^59 + 9*2^58 < 2^63).
//   * boundary data stays in halo2curves' R = 2^256 Montgomery form: mul29(X*2^256, W*2^261)
//     = X*W*2^256, so constants that multiply data (twiddles, SRS bases held by us) are stored
//     in R' form and data needs no conversion.
#pragma once
#include "ff.hip.hpp"

namespace zk {

constexpr uint32_t MASK29 = (1u << 29) - 1;

struct Fq29P {
    static constexpr uint32_t INV = 0x4866389u;
    __host__ __device__ static constexpr uint32_t M(int i) {
        constexpr uint32_t m[9] = {0x187cfd47u, 0x10460b6u, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu};
        return m[i];
    }
    using P32 = FqP;
};
struct Fr29P {
    static constexpr uint32_t INV = 0xfffffffu;
    __host__ __device__ static constexpr uint32_t M(int i) {
        constexpr uint32_t m[9] = {0x10000001u, 0x1f0fac9fu, 0xe5c2450u, 0x7d090f3u, 0x1585d283u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu};
        return m[i];
    }
    using P32 = FrP;
};

template <class P>
struct F29 {
    uint32_t l[9];
};

// 8 x 32 -> 9 x 29 (any 256-bit value)
template <class P>
__host__ __device__ __forceinline__ F29<P> unpack29(const Fp<typename P::P32>& a) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, w = bit >> 5, s = bit & 31;
        uint32_t v = a.l[w] >> s;
        if (s > 3 && w + 1 < 8) v |= a.l[w + 1] << (32 - s);
        r.l[i] = i == 8 ? v : (v & MASK29);
    }
    return r;
}
// carry-propagate so that limbs 0..7 < 2^29
template <class P>
__host__ __device__ __forceinline__ void normalize29(F29<P>& a) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a.l[i + 1] += a.l[i] >> 29; a.l[i] &= MASK29; }
}
// normalised 9 x 29 (value < 2^256) -> 8 x 32, no modular reduction
template <class P>
__host__ __device__ __forceinline__ Fp<typename P::P32> pack29_raw(const F29<P>& a) {
    Fp<typename P::P32> r;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int bit = 32 * w, i = bit / 29, s = bit - 29 * i;   // word w starts inside limb i at offset s
        uint32_t v = a.l[i] >> s;
        v |= a.l[i + 1] << (29 - s);
        if (29 - s + 29 < 32 && i + 2 < 9) v |= a.l[i + 2] << (58 - s);
        r.l[w] = v;
    }
    return r;
}
// full reduction to the canonical representative of a normalised value < 4p
template <class P>
__host__ __device__ __forceinline__ Fp<typename P::P32> pack29(const F29<P>& a) {
    Fp<typename P::P32> r = pack29_raw(a);
    cond_sub<typename P::P32>(r.l);   // value < 4p: subtract 2p? -> two conditional subtractions of p
    cond_sub<typename P::P32>(r.l);
    cond_sub<typename P::P32>(r.l);
    return r;
}

template <class P>
__host__ __device__ __forceinline__ F29<P> add29(const F29<P>& a, const F29<P>& b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// limb i of K*p in "balanced" form: normalised limbs n[i] of K*p with 2^29 borrowed from the next
// limb (c[i] = n[i] + 2^29 - [i > 0], c[8] = n[8] - 1), so c[i] >= 2^29 - 1 >= any normalised limb.
template <class P>
__host__ __device__ constexpr uint32_t kp_balanced(int K, int idx) {
    uint64_t carry = 0;
    uint32_t out = 0;
    for (int i = 0; i <= idx; ++i) {
        const uint64_t v = (uint64_t)K * P::M(i) + carry;
        out = i < 8 ? (uint32_t)(v & MASK29) : (uint32_t)v;
        carry = v >> 29;
    }
    if (idx < 8) out += 1u << 29;
    if (idx > 0) out -= 1u;
    return out;
}
// a - b + K*p, limb-wise non-negative for normalised b < K*p; result limbs < a.l[i] + 2^30.
// K*p must stay below the 2^261 capacity together with a (K <= 64 in practice).
template <int K, class P>
__host__ __device__ __forceinline__ F29<P> sub29k(const F29<P>& a, const F29<P>& b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + kp_balanced<P>(K, i) - b.l[i];
    return r;
}
// canonical representative of a normalised value < 2p
template <class P>
__host__ __device__ __forceinline__ Fp<typename P::P32> pack29_lt2p(const F29<P>& a) {
    Fp<typename P::P32> r = pack29_raw(a);
    cond_sub<typename P::P32>(r.l);
    return r;
}

// Montgomery product, product-scanning (FIPS) form, 64-bit column accumulator, no carries.
template <class P>
__host__ __device__ __forceinline__ F29<P> mul29(const F29<P>& a, const F29<P>& b) {
    uint32_t m[9];
    F29<P> t;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * P::M(k - i);
        m[k] = ((uint32_t)acc * P::INV) & MASK29;
        acc += (uint64_t)m[k] * P::M(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)m[i] * P::M(k - i);
        }
        t.l[k - 9] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    t.l[8] = (uint32_t)acc;
    return t;
}

// canonical representative of a normalised lazy value < 64 m (sums of a few dozen reduced terms):
// quotient estimate from the bits above 2^254 (both BN254 moduli are 0.756 * 2^254, so
// floor(t * 1354 / 1024) never exceeds floor(x / m) and misses it by at most 2), one multiple of m
// subtracted with signed carries, then conditional subtractions.  ~half the work of multiplying
// by one just to reduce.
template <class P>
__host__ __device__ __forceinline__ Fp<typename P::P32> reduce_lazy29(const F29<P>& x) {
    const uint32_t q = ((x.l[8] >> 22) * 1354u) >> 10;
    F29<P> r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int64_t s_ = (int64_t)x.l[i] - (int64_t)((uint64_t)q * P::M(i)) + c;
        r.l[i] = i < 8 ? (uint32_t)(s_ & MASK29) : (uint32_t)s_;
        c = s_ >> 29;
    }
    return pack29(r);          // r < 4m: up to three conditional subtractions
}

// Montgomery square: the cross products a_i a_j (i < j) are taken once against the doubled limb,
// 45 products instead of 81 in the operand part (the reduction part is unchanged): 126 vs 162.
// Same operand and result bounds as mul29(a, a).
template <class P>
__host__ __device__ __forceinline__ F29<P> sqr29(const F29<P>& a) {
    uint32_t m[9], a2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) a2[i] = a.l[i] << 1;
    F29<P> t;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; 2 * i < k; ++i) acc += (uint64_t)a2[i] * a.l[k - i];
        if (!(k & 1)) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * P::M(k - i);
        m[k] = ((uint32_t)acc * P::INV) & MASK29;
        acc += (uint64_t)m[k] * P::M(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; ++i) acc += (uint64_t)a2[i] * a.l[k - i];
        if (!(k & 1)) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * P::M(k - i);
        t.l[k - 9] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    t.l[8] = (uint32_t)acc;
    return t;
}

// a^(m-2) for a canonical 8 x 32 element in R = 2^256 Montgomery form; result in the same form.
// The 254-step exponentiation runs on 29-bit limbs (R' domain): its dependent chain is what a
// batch inversion waits for, and sqr29 / mul29 make it about half as long as the 32-bit CIOS one.
template <class P29>
__host__ __device__ inline Fp<typename P29::P32> inv_via29(const Fp<typename P29::P32>& a) {
    using F = Fp<typename P29::P32>;
    F t = a, one_rp = F::one();
#pragma unroll
    for (int i = 0; i < 5; ++i) { t = dbl(t); one_rp = dbl(one_rp); }     // x 32: R -> R'
    F29<P29> b = unpack29<P29>(t), r = unpack29<P29>(one_rp);
#pragma unroll 1
    for (int i = 0; i < 254; ++i) {
        const uint32_t limb = P29::P32::M(i >> 5) - (i < 32 ? 2u : 0u);     // bits of m - 2 (low limb >= 2: no borrow)
        if ((limb >> (i & 31)) & 1) r = mul29(r, b);
        b = sqr29(b);
    }
    return pack29_lt2p(mul29(r, unpack29<P29>(F::one())));                 // x 2^256 / 2^261: R' -> R
}

using Fq29 = F29<Fq29P>;
using Fr29 = F29<Fr29P>;

}  // namespace zk
