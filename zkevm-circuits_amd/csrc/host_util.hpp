// Host-side pieces of the halo2 prover surface that are inherently sequential and stay on the CPU
// ("transcript + RNG stay on the host"):
//   * Blake2b-512 (RFC 7693) with personalisation       -- blake2b_simd as used by halo2 transcripts
//   * (the transcripts built on it live in host_hash.hpp)
//   * XorShiftRng                                          -- rand_xorshift, the prover's RNG
//                                                             [REF prover/src/utils.rs:192-195]
//   * Fr helpers on 4 x u64 limbs (from_uniform_bytes, to_repr, random)
// Product code (not the test oracle).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/zkmi355.h"
#include "host_fq.hpp"

namespace zk {
namespace host {

// ------------------------------------------------------------------------------------ Blake2b
struct Blake2b {
    uint64_t h[8];
    uint64_t t0 = 0, t1 = 0;
    uint8_t buf[128];
    size_t buflen = 0;

    static constexpr uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    static inline uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

    // digest length 64, no key, 16-byte personalisation (zero padded)
    void init(const char* personal) {
        uint8_t p[64];
        memset(p, 0, sizeof p);
        p[0] = 64;   // digest length
        p[2] = 1;    // fanout
        p[3] = 1;    // depth
        if (personal) { size_t l = strlen(personal); if (l > 16) l = 16; memcpy(p + 48, personal, l); }
        for (int i = 0; i < 8; ++i) { uint64_t w; memcpy(&w, p + 8 * i, 8); h[i] = IV[i] ^ w; }
        t0 = t1 = 0;
        buflen = 0;
    }
    void compress(const uint8_t* block, bool last) {
        static const uint8_t S[12][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        uint64_t m[16], v[16];
        memcpy(m, block, 128);
        for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = IV[i]; }
        v[12] ^= t0;
        v[13] ^= t1;
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; ++r) {
            const uint8_t* s = S[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
    }
    void update(const void* data, size_t len) {
        const uint8_t* in = (const uint8_t*)data;
        while (len) {
            if (buflen == 128) {       // buffer full and more input follows: not the last block
                t0 += 128; if (t0 < 128) ++t1;
                compress(buf, false);
                buflen = 0;
            }
            size_t take = 128 - buflen;
            if (take > len) take = len;
            memcpy(buf + buflen, in, take);
            buflen += take; in += take; len -= take;
        }
    }
    // non-destructive: works on a copy so the transcript can keep absorbing
    void finalize(uint8_t out[64]) const {
        Blake2b c = *this;
        c.t0 += c.buflen; if (c.t0 < c.buflen) ++c.t1;
        memset(c.buf + c.buflen, 0, 128 - c.buflen);
        c.compress(c.buf, true);
        memcpy(out, c.h, 64);
    }
};

// ------------------------------------------------------------------------------------ Fr (host)
struct FrK {   // extra Fr constants: R^2 mod r
    static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
};
inline F4 fr_add(const F4& a, const F4& b) { return fadd<FrC>(a, b); }
inline F4 fr_sub(const F4& a, const F4& b) { return fsub<FrC>(a, b); }
inline F4 fr_mul(const F4& a, const F4& b) { return fmul<FrC>(a, b); }
inline F4 fr_inv(const F4& a) { return finv<FrC>(a); }
inline F4 fr_one() { return fone<FrC>(); }
inline F4 fr_zero() { F4 z; memset(&z, 0, sizeof z); return z; }
inline bool fr_is_zero(const F4& a) { return fzero<FrC>(a); }
inline bool fr_eq(const F4& a, const F4& b) { return memcmp(&a, &b, 32) == 0; }
inline F4 fr_r2() { F4 r; memcpy(r.l, FrK::R2, 32); return r; }
// plain 256-bit integer (any value < 2^256) -> Montgomery form of (value mod r)
inline F4 fr_to_mont(const F4& a) { return fr_mul(a, fr_r2()); }
inline F4 fr_from_u64(uint64_t v) { F4 a = fr_zero(); a.l[0] = v; return fr_to_mont(a); }
// Montgomery form -> canonical integer
inline F4 fr_canon(const F4& a) { F4 one = fr_zero(); one.l[0] = 1; return fr_mul(a, one); }
inline void fr_to_repr(const F4& a, uint8_t out[32]) { F4 c = fr_canon(a); memcpy(out, c.l, 32); }
// Fr::from_uniform_bytes: 64 little-endian bytes mod r  (golden G3 pins this in the oracle)
inline F4 fr_from_uniform(const uint8_t b[64]) {
    F4 lo, hi;
    memcpy(lo.l, b, 32);
    memcpy(hi.l, b + 32, 32);
    return fr_add(fr_to_mont(lo), fr_mul(fr_to_mont(hi), fr_r2()));   // lo + hi * 2^256
}
inline F4 fr_pow(F4 base, uint64_t e) {
    F4 r = fr_one();
    while (e) { if (e & 1) r = fr_mul(r, base); base = fr_mul(base, base); e >>= 1; }
    return r;
}

// ------------------------------------------------------------------------------------ RNG
struct XorShiftRng {   // rand_xorshift 0.3
    uint32_t x, y, z, w;
    explicit XorShiftRng(const uint8_t seed[16]) {
        uint32_t s[4];
        memcpy(s, seed, 16);
        if ((s[0] | s[1] | s[2] | s[3]) == 0) { s[0] = 0x193a6754u; s[1] = 0xa8a7d469u; s[2] = 0x97830e05u; s[3] = 0x113ba7bbu; }
        x = s[0]; y = s[1]; z = s[2]; w = s[3];
    }
    uint32_t next_u32() {
        const uint32_t t = x ^ (x << 11);
        x = y; y = z; z = w;
        w = w ^ (w >> 19) ^ (t ^ (t >> 8));
        return w;
    }
    uint64_t next_u64() { const uint64_t a = next_u32(), b = next_u32(); return (b << 32) | a; }
    // halo2curves Fr::random: from_u512 of eight next_u64 words
    F4 next_fr() {
        uint8_t b[64];
        for (int i = 0; i < 8; ++i) { const uint64_t v = next_u64(); memcpy(b + 8 * i, &v, 8); }
        return fr_from_uniform(b);
    }
};

// ------------------------------------------------------------------------------------ points
inline void fq_to_repr(const Fq& a, uint8_t out[32]) {
    F4 m, one;
    memcpy(m.l, &a, 32);
    memset(&one, 0, sizeof one);
    one.l[0] = 1;
    F4 c = fmul<FqC>(m, one);
    memcpy(out, c.l, 32);
}
// halo2curves @ a495a7b compressed G1 (`new_curve_impl!`, the flags live in the two spare top bits of the last byte): x canonical
// LE, bit 254 (0x40 of byte 31) = parity of y, bit 255 (0x80) = identity (with a zero x).  Pinned by the reference's own vk / proof
// bytes  [REF aggregator/data/batch-task.json: chunk_proofs[0]]  (tests/test_reference_chunk_proof.py): 18 of 18 points carry
// (y & 1) << 6 and never bit 7.  The identity's image is the same macro's, not readable off that fixture.
constexpr uint8_t G1_FLAG_SIGN = 0x40, G1_FLAG_IDENTITY = 0x80;
inline void g1_compress(const G1Affine& p, uint8_t out[32]) {
    if (p.is_identity()) { memset(out, 0, 32); out[31] = G1_FLAG_IDENTITY; return; }
    uint8_t y[32];
    fq_to_repr(p.x, out);
    fq_to_repr(p.y, y);
    out[31] |= (uint8_t)((y[0] & 1) << 6);
}

}  // namespace host
}  // namespace zk
