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
// The same for a subtrahend whose limbs are plain sums of S normalised limbs (b.l[i] <= S * (2^29 - 1), b < K*p - S * 2^232):
// S borrows of 2^29 per limb instead of one.  Result limbs < a.l[i] + (S + 1) * 2^29.
template <int K, int S, class P>
__host__ __device__ __forceinline__ F29<P> sub29kw(const F29<P>& a, const F29<P>& b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint32_t c = kp_balanced<P>(K, i);
        if (i < 8) c += (uint32_t)(S - 1) << 29;
        if (i > 0) c -= (uint32_t)(S - 1);
        r.l[i] = a.l[i] + c - b.l[i];
    }
    return r;
}
// canonical representative of a normalised value < 2p
template <class P>
__host__ __device__ __forceinline__ Fp<typename P::P32> pack29_lt2p(const F29<P>& a) {
    Fp<typename P::P32> r = pack29_raw(a);
    cond_sub<typename P::P32>(r.l);
    return r;
}

// Column sums of the products below: acc + sum_j x[j]*y[j] (dotv: both factors in registers) and acc + sum_j x[j]*k[j]
// (dotk: k[] compile-time constants, held in scalar registers).  Left to itself the compiler starts every column sum from
// zero (so that the multiplications do not wait for the carry out of the previous column) and joins the carry with a 64-bit
// add of its own: 16 instructions per product, bought for latency that four resident waves hide anyway.  ZK_MAD_CHAIN makes
// the multiply-adds opaque so that the sums run as written, carry first: 1 = one asm statement per multiply-add, 2 = one
// per column part (the hazard recogniser puts an s_nop behind every asm statement).  0 = plain C, the association is the
// compiler's; host compilation always takes this form.  Measured (tools/gpu_arith_ab.sh, one MI355X): mode 2 against mode 0
// k_msm_buckets 0.998 -> 0.983 ms, the weighted bucket sum chain 2.01 -> 1.77 ms, NTT 2^20 124 -> 120 us, evaluator 3.43 ->
// 3.34 ms; it also needs fewer registers (the evaluator 95 -> 77, the NTT passes lose their scratch spills).
#ifndef ZK_MAD_CHAIN
#define ZK_MAD_CHAIN 2
#endif
#if ZK_MAD_CHAIN && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint64_t dotv_asm(const uint32_t* x, const uint32_t* y, int n, uint64_t acc) {
    switch (n) {
        case 1:
            asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(y[0]) : "vcc");
            break;
        case 2:
            asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %4, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(y[0]), "v"(y[1]) : "vcc");
            break;
        case 3:
            asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %5, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %6, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(y[0]), "v"(y[1]), "v"(y[2]) : "vcc");
            break;
        case 4:
            asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %6, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %7, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %8, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]) : "vcc");
            break;
        case 5:
            asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %7, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %8, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %9, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %5, %10, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]) : "vcc");
            break;
        case 6:
            asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %8, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %9, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %10, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %5, %11, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %6, %12, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]) : "vcc");
            break;
        case 7:
            asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %9, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %10, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %11, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %5, %12, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %6, %13, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %7, %14, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]) : "vcc");
            break;
        case 8:
            asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %10, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %11, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %12, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %5, %13, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %6, %14, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %7, %15, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %8, %16, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]) : "vcc");
            break;
        case 9:
            asm("v_mad_u64_u32 %0, vcc, %1, %10, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %11, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %12, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %13, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %5, %14, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %6, %15, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %7, %16, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %8, %17, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %9, %18, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8]) : "vcc");
            break;
        default: break;
    }
    return acc;
}
__device__ __forceinline__ uint64_t dotk_asm(const uint32_t* x, const uint32_t* y, int n, uint64_t acc) {
    switch (n) {
        case 1:
            asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "s"(y[0]) : "vcc");
            break;
        case 2:
            asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %4, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "s"(y[0]), "s"(y[1]) : "vcc");
            break;
        case 3:
            asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %5, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %6, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "s"(y[0]), "s"(y[1]), "s"(y[2]) : "vcc");
            break;
        case 4:
            asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %6, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %7, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %8, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "s"(y[0]), "s"(y[1]), "s"(y[2]), "s"(y[3]) : "vcc");
            break;
        case 5:
            asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %7, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %8, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %9, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %5, %10, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "s"(y[0]), "s"(y[1]), "s"(y[2]), "s"(y[3]), "s"(y[4]) : "vcc");
            break;
        case 6:
            asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %8, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %9, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %10, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %5, %11, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %6, %12, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "s"(y[0]), "s"(y[1]), "s"(y[2]), "s"(y[3]), "s"(y[4]), "s"(y[5]) : "vcc");
            break;
        case 7:
            asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %9, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %10, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %11, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %5, %12, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %6, %13, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %7, %14, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "s"(y[0]), "s"(y[1]), "s"(y[2]), "s"(y[3]), "s"(y[4]), "s"(y[5]), "s"(y[6]) : "vcc");
            break;
        case 8:
            asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %10, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %11, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %12, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %5, %13, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %6, %14, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %7, %15, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %8, %16, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "s"(y[0]), "s"(y[1]), "s"(y[2]), "s"(y[3]), "s"(y[4]), "s"(y[5]), "s"(y[6]), "s"(y[7]) : "vcc");
            break;
        case 9:
            asm("v_mad_u64_u32 %0, vcc, %1, %10, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %11, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %12, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %13, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %5, %14, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %6, %15, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %7, %16, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %8, %17, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %9, %18, %0\n\t"
                : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "s"(y[0]), "s"(y[1]), "s"(y[2]), "s"(y[3]), "s"(y[4]), "s"(y[5]), "s"(y[6]), "s"(y[7]), "s"(y[8]) : "vcc");
            break;
        default: break;
    }
    return acc;
}
#endif
__host__ __device__ __forceinline__ uint64_t dotv(const uint32_t* x, const uint32_t* y, int n, uint64_t acc) {
#if ZK_MAD_CHAIN == 2 && defined(__HIP_DEVICE_COMPILE__)
    return dotv_asm(x, y, n, acc);
#elif ZK_MAD_CHAIN == 1 && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int j = 0; j < n; ++j) acc = dotv_asm(x + j, y + j, 1, acc);
    return acc;
#else
#pragma unroll
    for (int j = 0; j < n; ++j) acc += (uint64_t)x[j] * y[j];
    return acc;
#endif
}
__host__ __device__ __forceinline__ uint64_t dotk(const uint32_t* x, const uint32_t* k, int n, uint64_t acc) {
#if ZK_MAD_CHAIN == 2 && defined(__HIP_DEVICE_COMPILE__)
    return dotk_asm(x, k, n, acc);
#elif ZK_MAD_CHAIN == 1 && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int j = 0; j < n; ++j) acc = dotk_asm(x + j, k + j, 1, acc);
    return acc;
#else
#pragma unroll
    for (int j = 0; j < n; ++j) acc += (uint64_t)x[j] * k[j];
    return acc;
#endif
}

// Montgomery product, product-scanning (FIPS) form, 64-bit column accumulator, no carries.
template <class P>
__host__ __device__ __forceinline__ F29<P> mul29_c(const F29<P>& a, const F29<P>& b) {
    uint32_t m[9];
    F29<P> t;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; ++k) {
        // column k: a[i] * b[k - i] and m[i] * M[k - i] over lo <= i <= hi (the m[k] * M[0] term of k < 9 comes once m[k] is known)
        const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8, nm = k < 9 ? k : 17 - k;
        uint32_t ys[9], ks[9];
#pragma unroll
        for (int i = lo; i <= hi; ++i) { ys[i - lo] = b.l[k - i]; ks[i - lo] = P::M(k - i); }
        acc = dotv(a.l + lo, ys, hi - lo + 1, acc);
        acc = dotk(m + lo, ks, nm, acc);
        if (k < 9) {
            m[k] = ((uint32_t)acc * P::INV) & MASK29;
            acc += (uint64_t)m[k] * P::M(0);
        } else {
            t.l[k - 9] = (uint32_t)acc & MASK29;
        }
        acc >>= 29;
    }
    t.l[8] = (uint32_t)acc;
    return t;
}

// The same product with a wave-uniform second factor (a constant of a program, a challenge): its limbs stay in scalar
// registers and enter the multiply-adds as the scalar operand, like the modulus -- no vector copies of b.
template <class P>
__host__ __device__ __forceinline__ F29<P> mul29_ub_c(const F29<P>& a, const F29<P>& b) {
    uint32_t m[9];
    F29<P> t;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; ++k) {
        const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8, nm = k < 9 ? k : 17 - k;
        uint32_t ys[9], ks[9];
#pragma unroll
        for (int i = lo; i <= hi; ++i) { ys[i - lo] = b.l[k - i]; ks[i - lo] = P::M(k - i); }
        acc = dotk(a.l + lo, ys, hi - lo + 1, acc);
        acc = dotk(m + lo, ks, nm, acc);
        if (k < 9) {
            m[k] = ((uint32_t)acc * P::INV) & MASK29;
            acc += (uint64_t)m[k] * P::M(0);
        } else {
            t.l[k - 9] = (uint32_t)acc & MASK29;
        }
        acc >>= 29;
    }
    t.l[8] = (uint32_t)acc;
    return t;
}

// (a*b + c*d) * 2^-261 mod p in one pass: the second product joins the column sums of the first, one reduction serves both
// (243 v_mad_u64_u32 instead of 324 + the subtraction or addition that would have combined two reduced products).
// a, b, c normalised (limbs < 2^29), d with limbs < 2^30 (e.g. K*p - x taken limb-wise, see neg29k), a*b + c*d < 2^261 * p:
// a column holds at most 9*2^58 + 9*2^59 + 9*2^58 < 2^64.  Result normalised, < (a*b + c*d) / 2^261 + p.
template <class P>
__host__ __device__ __forceinline__ F29<P> mul2add29_c(const F29<P>& a, const F29<P>& b, const F29<P>& c, const F29<P>& d) {
    uint32_t m[9];
    F29<P> t;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; ++k) {
        const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8, nm = k < 9 ? k : 17 - k;
        uint32_t ys[9], ds[9], ks[9];
#pragma unroll
        for (int i = lo; i <= hi; ++i) { ys[i - lo] = b.l[k - i]; ds[i - lo] = d.l[k - i]; ks[i - lo] = P::M(k - i); }
        acc = dotv(a.l + lo, ys, hi - lo + 1, acc);
        acc = dotv(c.l + lo, ds, hi - lo + 1, acc);
        acc = dotk(m + lo, ks, nm, acc);
        if (k < 9) {
            m[k] = ((uint32_t)acc * P::INV) & MASK29;
            acc += (uint64_t)m[k] * P::M(0);
        } else {
            t.l[k - 9] = (uint32_t)acc & MASK29;
        }
        acc >>= 29;
    }
    t.l[8] = (uint32_t)acc;
    return t;
}
// K*p - b taken limb by limb (b normalised, top limb of b below the top limb of K*p, i.e. b < K*p - 2^232): the value is
// K*p - b, limbs are non-negative and below 2^30, not normalised -- what mul2add29 takes as its last operand.
template <int K, class P>
__host__ __device__ __forceinline__ F29<P> neg29k(const F29<P>& b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = kp_balanced<P>(K, i) - b.l[i];
    return r;
}

// canonical representative of a normalised lazy value < 64 m (sums of a few dozen reduced terms): quotient estimate from the bits
// above 2^232 -- t = x >> 232 is the top limb, q = floor(t C / 2^52) with C = floor(2^92 / ((m >> 192) + 1)) <= 2^284 / m (31 bits; both
// BN254 moduli share their top 128 bits) -- never exceeds x / m and misses it by less than 1 + 2^-20, so x - q m lies in
// [0, (1 + 2^-20) m): one multiple of m subtracted with signed carries, then ONE conditional subtraction (round 5; the estimate from
// six bits that this replaces missed by up to 2 and paid three conditional subtractions -- 50 instructions per output of the NTT's
// last pass).  ~half the work of multiplying by one just to reduce.
template <class P>
__host__ __device__ __forceinline__ Fp<typename P::P32> reduce_lazy29(const F29<P>& x) {
    constexpr uint64_t mhi = ((uint64_t)P::P32::M(7) << 32) | P::P32::M(6);
    constexpr uint32_t C = (uint32_t)((((unsigned __int128)1) << 92) / ((unsigned __int128)mhi + 1));
    const uint32_t q = (uint32_t)(((uint64_t)x.l[8] * C) >> 52);
    F29<P> r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int64_t s_ = (int64_t)x.l[i] - (int64_t)((uint64_t)q * P::M(i)) + c;
        r.l[i] = i < 8 ? (uint32_t)(s_ & MASK29) : (uint32_t)s_;
        c = s_ >> 29;
    }
    return pack29_lt2p(r);          // r < (1 + 2^-20) m
}

// Montgomery square: the cross products a_i a_j (i < j) are taken once against the doubled limb,
// 45 products instead of 81 in the operand part (the reduction part is unchanged): 126 vs 162.
// Same operand and result bounds as mul29(a, a).
template <class P>
__host__ __device__ __forceinline__ F29<P> sqr29_c(const F29<P>& a) {
    uint32_t m[9], a2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) a2[i] = a.l[i] << 1;
    F29<P> t;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; ++k) {
        const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8, nm = k < 9 ? k : 17 - k;
        // cross terms lo <= i < k - i, then the square of the middle limb of an even column
        uint32_t xs[9], ys[9], ks[9];
        int n = 0;
#pragma unroll
        for (int i = lo; 2 * i < k; ++i) { xs[n] = a2[i]; ys[n] = a.l[k - i]; ++n; }
        if (!(k & 1)) { xs[n] = a.l[k / 2]; ys[n] = a.l[k / 2]; ++n; }
#pragma unroll
        for (int i = lo; i <= hi; ++i) ks[i - lo] = P::M(k - i);
        acc = dotv(xs, ys, n, acc);
        acc = dotk(m + lo, ks, nm, acc);
        if (k < 9) {
            m[k] = ((uint32_t)acc * P::INV) & MASK29;
            acc += (uint64_t)m[k] * P::M(0);
        } else {
            t.l[k - 9] = (uint32_t)acc & MASK29;
        }
        acc >>= 29;
    }
    t.l[8] = (uint32_t)acc;
    return t;
}


// The products as the rest of the library calls them.  On the device (ZK_MUL_ASM, default 1) each is ONE asm statement generated by
// tools/gen_mul29_asm.py (csrc/mul29_asm.hip.hpp) -- the same column sums in the same order as the C forms above, which stay the
// definition (host compilation, tests/test_host_arith.py) and what the device forms are compared with on the GPU (tools/ubench.hip:
// 4 M operand sets at the documented bounds, bit-identical).  With one statement per column part (ZK_MAD_CHAIN = 2, the C forms) a
// product carries its 162 multiply-adds in ~260 instructions; written out whole it is 162 + 9 v_mul_lo + 26 v_and + 17 v_lshrrev_b64 + 1.
// The bare product chain does not gain from that (179 vs 176 G/s: its other instructions hide behind the multiply-adds), the kernels
// do: NTT class of the headline proof 718 -> 700 ms, proof 1.033 -> 1.014 s on one box (alternating A/B, profiles/r05_experiments.md).
// `make VARIANT=c EXTRA=-DZK_MUL_ASM=0` builds the C forms for A/B runs.
#ifndef ZK_MUL_ASM
#define ZK_MUL_ASM 1
#endif
#if ZK_MUL_ASM && defined(__HIP_DEVICE_COMPILE__)
#include "mul29_asm.hip.hpp"
#define ZK_MUL29_DISPATCH(NAME, ...) return NAME##_asm<P>(__VA_ARGS__)
#else
#define ZK_MUL29_DISPATCH(NAME, ...) return NAME##_c<P>(__VA_ARGS__)
#endif
template <class P>
__host__ __device__ __forceinline__ F29<P> mul29(const F29<P>& a, const F29<P>& b) { ZK_MUL29_DISPATCH(mul29, a, b); }
template <class P>
__host__ __device__ __forceinline__ F29<P> mul29_ub(const F29<P>& a, const F29<P>& b) { ZK_MUL29_DISPATCH(mul29_ub, a, b); }
template <class P>
__host__ __device__ __forceinline__ F29<P> mul2add29(const F29<P>& a, const F29<P>& b, const F29<P>& c, const F29<P>& d) { ZK_MUL29_DISPATCH(mul2add29, a, b, c, d); }
template <class P>
__host__ __device__ __forceinline__ F29<P> sqr29(const F29<P>& a) { ZK_MUL29_DISPATCH(sqr29, a); }
#undef ZK_MUL29_DISPATCH
// In-place forms (round 6; the evaluator's interpreter keeps its top of stack in fixed registers): on the device the result is
// written over the named factor's registers by the asm statement itself -- no copy, no phi at the join of an interpreter's switch.
//   mul29_ipa(x, y): x <- mul29(x, y)      mul29_ipb(x, y): x <- mul29(y, x)      mul29_ub_ipa(x, y): x <- mul29_ub(x, y)
template <class P>
__host__ __device__ __forceinline__ void mul29_ipa(F29<P>& x, const F29<P>& y) {
#if ZK_MUL_ASM && defined(__HIP_DEVICE_COMPILE__)
    mul29_ipa_asm<P>(x, y);
#else
    x = mul29_c<P>(x, y);
#endif
}
template <class P>
__host__ __device__ __forceinline__ void mul29_ipb(F29<P>& x, const F29<P>& y) {
#if ZK_MUL_ASM && defined(__HIP_DEVICE_COMPILE__)
    mul29_ipb_asm<P>(x, y);
#else
    x = mul29_c<P>(y, x);
#endif
}
template <class P>
__host__ __device__ __forceinline__ void mul29_ub_ipa(F29<P>& x, const F29<P>& y) {
#if ZK_MUL_ASM && defined(__HIP_DEVICE_COMPILE__)
    mul29_ub_ipa_asm<P>(x, y);
#else
    x = mul29_ub_c<P>(x, y);
#endif
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
