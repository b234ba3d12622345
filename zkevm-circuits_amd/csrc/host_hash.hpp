// Host-side hash functions of the reference's non-Blake2b transcripts (transcripts
// are sequential and stay on the CPU).  Product code, not the test oracle.
//   * Keccak-256 (Keccak padding 0x01, the EVM's KECCAK256)  -- snark-verifier EvmTranscript, reached
//     from gen_evm_proof_shplonk [REF prover/src/common/prover/evm.rs:67]
//   * Poseidon over BN254 Fr, x^5, T = 5, RATE = 4, R_F = 8, R_P = 60 (POSEIDON_SPEC of
//     snark-verifier-sdk [REF aggregator/src/core.rs:25-28,57-58]) with the sponge discipline of
//     snark-verifier's util::hash::Poseidon (buffered update, state[1] as the squeeze output)
//     -- PoseidonTranscript<NativeLoader, _>, reached from gen_snark_shplonk
//     [REF prover/src/common/prover/utils.rs:31]
// Round constants and the MDS matrix come from the Grain LFSR of the Poseidon paper (eprint
// 2019/458, suppl. F) exactly as the `poseidon` crate derives them; the plain round schedule used
// here computes the same permutation as that crate's sparse-matrix form.  tests/ pins both hashes
// through the oracle (keccak256("") held by the reference, poseidonperm_x5_254_5).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "host_util.hpp"

namespace zk {
namespace host {

// ------------------------------------------------------------------------------------ Keccak-256
inline void keccak_f1600(uint64_t a[25]) {
    static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL, 0x0000000080000001ULL,
                                    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
                                    0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
                                    0x000000000000800AULL, 0x800000008000000AULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};   // [x + 5 y]
    auto rol = [](uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; };
    for (int rnd = 0; rnd < 24; ++rnd) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y], ROT[x + 5 * y]);
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[rnd];
    }
}
inline void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
    const size_t rate = 136;
    uint64_t a[25];
    memset(a, 0, sizeof a);
    uint8_t block[136];
    while (true) {
        const size_t take = len < rate ? len : rate;
        memset(block, 0, rate);
        memcpy(block, data, take);
        const bool last = take < rate;
        if (last) { block[take] ^= 0x01; block[rate - 1] ^= 0x80; }
        for (size_t i = 0; i < rate / 8; ++i) { uint64_t w; memcpy(&w, block + 8 * i, 8); a[i] ^= w; }
        keccak_f1600(a);
        data += take; len -= take;
        if (last) break;
    }
    memcpy(out, a, 32);
}

// ------------------------------------------------------------------------------------ Poseidon
// One parameter set (x^5 over BN254 Fr): round constants and Cauchy MDS matrix from the Grain LFSR, the plain round schedule.
// Two instances are in use: <5, 8, 60> = POSEIDON_SPEC, the transcript's permutation; <3, 8, 57> = the width of the reference's
// Poseidon code hash, whose value for empty code (`POSEIDON_CODE_HASH_EMPTY`, eth-types/src/lib.rs:278) is permute(0, 0, 0)[0] --
// a vector held by the reference that pins this generator.
template <int T_, int R_F_, int R_P_>
struct PoseidonPerm {
    static constexpr int T = T_, RATE = T_ - 1, R_F = R_F_, R_P = R_P_;
    std::vector<F4> constants;      // [(R_F + R_P)][T]
    F4 mds[T][T];

    struct Grain {
        uint8_t s[80];
        int head = 0;               // ring buffer: bit i of the register is s[(head + i) % 80]
        uint8_t at(int i) const { return s[(head + i) % 80]; }
        uint8_t new_bit() {
            const uint8_t nb = at(62) ^ at(51) ^ at(38) ^ at(23) ^ at(13) ^ at(0);
            s[head] = nb;           // the oldest bit leaves, the new one takes the last position
            head = (head + 1) % 80;
            return nb;
        }
        uint8_t next_bit() {
            while (!new_bit()) new_bit();       // a 0 discards the bit that follows it
            return new_bit();
        }
        void init(uint32_t field_bits, uint32_t t, uint32_t r_f, uint32_t r_p) {
            int pos = 0;
            auto app = [&](int n, uint32_t v) { for (int i = 0; i < n; ++i) s[pos++] = (uint8_t)((v >> (n - 1 - i)) & 1u); };
            app(2, 1); app(4, 0); app(12, field_bits); app(12, t); app(10, r_f); app(10, r_p); app(30, 0x3FFFFFFFu);
            head = 0;
            for (int i = 0; i < 160; ++i) new_bit();
        }
        // 254 bits, most significant first, as a plain 256-bit integer
        F4 next_int254() {
            F4 v = fr_zero();
            for (int i = 253; i >= 0; --i) if (next_bit()) v.l[i / 64] |= 1ull << (i % 64);
            return v;
        }
    };
    static bool lt_modulus(const F4& v) {
        static const uint64_t M[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
        for (int i = 3; i >= 0; --i) { if (v.l[i] < M[i]) return true; if (v.l[i] > M[i]) return false; }
        return false;
    }
    PoseidonPerm() {
        Grain g;
        g.init(254, T, R_F, R_P);
        constants.resize((size_t)(R_F + R_P) * T);
        for (F4& c : constants) {
            F4 v;
            do v = g.next_int254(); while (!lt_modulus(v));      // rejection sampling
            c = fr_to_mont(v);
        }
        F4 xs[T], ys[T];
        for (int i = 0; i < T; ++i) xs[i] = fr_to_mont(g.next_int254());           // reduced, not rejected
        for (int i = 0; i < T; ++i) ys[i] = fr_to_mont(g.next_int254());
        for (int i = 0; i < T; ++i)
            for (int j = 0; j < T; ++j) mds[i][j] = fr_inv(fr_add(xs[i], ys[j]));  // Cauchy matrix
    }
    static const PoseidonPerm& get() { static const PoseidonPerm spec; return spec; }

    static F4 pow5(const F4& v) { const F4 v2 = fr_mul(v, v); return fr_mul(fr_mul(v2, v2), v); }
    void permute(F4 s[T]) const {
        const int half = R_F / 2;
        for (int rnd = 0; rnd < R_F + R_P; ++rnd) {
            for (int i = 0; i < T; ++i) s[i] = fr_add(s[i], constants[(size_t)rnd * T + i]);
            if (rnd < half || rnd >= half + R_P) { for (int i = 0; i < T; ++i) s[i] = pow5(s[i]); }
            else s[0] = pow5(s[0]);
            F4 o[T];
            for (int i = 0; i < T; ++i) {
                F4 acc = fr_mul(mds[i][0], s[0]);
                for (int j = 1; j < T; ++j) acc = fr_add(acc, fr_mul(mds[i][j], s[j]));
                o[i] = acc;
            }
            for (int i = 0; i < T; ++i) s[i] = o[i];
        }
    }
};

using PoseidonSpec = PoseidonPerm<5, 8, 60>;
using PoseidonWidth3 = PoseidonPerm<3, 8, 57>;

// snark-verifier util::hash::Poseidon: `update` buffers; `squeeze` absorbs RATE elements per
// permutation (a short chunk -- or an extra empty one after an exact multiple -- gets the padding
// element 1 behind its last input) and returns state[1].  Initial state (2^64, 0, 0, 0, 0).
struct PoseidonSponge {
    F4 state[PoseidonSpec::T];
    std::vector<F4> buf;
    PoseidonSponge() {
        for (F4& v : state) v = fr_zero();
        F4 cap = fr_zero();
        cap.l[1] = 1;                       // 2^64
        state[0] = fr_to_mont(cap);
    }
    void update(const F4& e) { buf.push_back(e); }
    void absorb(const F4* chunk, size_t len) {
        for (size_t i = 0; i < len; ++i) state[1 + i] = fr_add(state[1 + i], chunk[i]);
        if (len < (size_t)PoseidonSpec::RATE) state[1 + len] = fr_add(state[1 + len], fr_one());
        PoseidonSpec::get().permute(state);
    }
    F4 squeeze() {
        const size_t rate = PoseidonSpec::RATE;
        const bool exact = buf.size() % rate == 0;
        for (size_t off = 0; off < buf.size(); off += rate) absorb(buf.data() + off, std::min(rate, buf.size() - off));
        if (exact) absorb(nullptr, 0);
        buf.clear();
        return state[1];
    }
};

// ------------------------------------------------------------------------------------ transcripts
// halo2 Transcript / TranscriptWrite, three built-in kinds plus forwarding to the caller's object:
//   ZK_TRANSCRIPT_BLAKE2B   halo2_proofs::transcript::Blake2bWrite<_, G1Affine, Challenge255<_>>
//   ZK_TRANSCRIPT_POSEIDON  snark-verifier PoseidonTranscript<G1Affine, NativeLoader, _> (coordinates enter
//                           as Fq integers reduced mod r; points written compressed, scalars little-endian)
//   ZK_TRANSCRIPT_EVM       snark-verifier EvmTranscript (Keccak-256 over big-endian words; points written
//                           as x || y, 64 bytes; a lone 32-byte state gets a 0x01 suffix before re-hashing)
// With an external vtable (zk_proof_set_transcript) every operation is forwarded to the host
// language's own transcript object, which then also owns the proof bytes.
inline void fq_to_be(const Fq& a, uint8_t out[32]) {
    uint8_t le[32];
    fq_to_repr(a, le);
    for (int i = 0; i < 32; ++i) out[i] = le[31 - i];
}
struct Transcript {
    int kind = ZK_TRANSCRIPT_BLAKE2B;
    Blake2b st;
    PoseidonSponge sponge;
    std::vector<uint8_t> evm;
    std::vector<uint8_t> proof;
    const zk_transcript_vtable* vt = nullptr;
    void* user = nullptr;
    int err = 0;                              // first non-zero status: external callback failure, or a point the kind cannot absorb
    Transcript() { st.init("Halo2-Transcript"); }
    void reset(int k) { kind = k; st.init("Halo2-Transcript"); sponge = PoseidonSponge(); evm.clear(); proof.clear(); err = 0; }
    void common_point(const G1Affine& p) {
        if (vt) { if (int rc = vt->common_point(user, &p)) err = err ? err : rc; return; }
        if (kind == ZK_TRANSCRIPT_BLAKE2B) {
            uint8_t b[65];
            b[0] = 1;   // BLAKE2B_PREFIX_POINT
            if (p.is_identity()) memset(b + 1, 0, 64);
            else { fq_to_repr(p.x, b + 1); fq_to_repr(p.y, b + 33); }
            st.update(b, 65);
            return;
        }
        if (p.is_identity()) { err = err ? err : ZK_ERR_INVALID_ARG; return; }     // snark-verifier: the identity has no coordinates
        if (kind == ZK_TRANSCRIPT_POSEIDON) {
            F4 x, y;
            fq_to_repr(p.x, (uint8_t*)x.l);
            fq_to_repr(p.y, (uint8_t*)y.l);
            sponge.update(fr_to_mont(x));       // fe_to_fe: the base-field integer mod r
            sponge.update(fr_to_mont(y));
        } else {
            uint8_t b[64];
            fq_to_be(p.x, b);
            fq_to_be(p.y, b + 32);
            evm.insert(evm.end(), b, b + 64);
        }
    }
    void common_scalar(const F4& s) {
        if (vt) { if (int rc = vt->common_scalar(user, &s)) err = err ? err : rc; return; }
        if (kind == ZK_TRANSCRIPT_BLAKE2B) {
            uint8_t b[33];
            b[0] = 2;   // BLAKE2B_PREFIX_SCALAR
            fr_to_repr(s, b + 1);
            st.update(b, 33);
        } else if (kind == ZK_TRANSCRIPT_POSEIDON) {
            sponge.update(s);
        } else {
            uint8_t le[32];
            fr_to_repr(s, le);
            for (int i = 0; i < 32; ++i) evm.push_back(le[31 - i]);
        }
    }
    void write_point(const G1Affine& p) {
        if (vt) { if (int rc = vt->write_point(user, &p)) err = err ? err : rc; return; }
        common_point(p);
        if (kind == ZK_TRANSCRIPT_EVM) {
            uint8_t b[64];
            fq_to_be(p.x, b);
            fq_to_be(p.y, b + 32);
            proof.insert(proof.end(), b, b + 64);
        } else {
            uint8_t c[32];
            g1_compress(p, c);
            proof.insert(proof.end(), c, c + 32);
        }
    }
    void write_scalar(const F4& s) {
        if (vt) { if (int rc = vt->write_scalar(user, &s)) err = err ? err : rc; return; }
        common_scalar(s);
        uint8_t c[32];
        fr_to_repr(s, c);
        if (kind == ZK_TRANSCRIPT_EVM) for (int i = 31; i >= 0; --i) proof.push_back(c[i]);
        else proof.insert(proof.end(), c, c + 32);
    }
    F4 squeeze() {
        if (vt) {
            F4 out = fr_zero();
            if (int rc = vt->squeeze_challenge(user, &out)) err = err ? err : rc;
            return out;
        }
        if (kind == ZK_TRANSCRIPT_POSEIDON) return sponge.squeeze();
        if (kind == ZK_TRANSCRIPT_EVM) {
            if (evm.size() == 32) evm.push_back(1);
            uint8_t h[32];
            keccak256(evm.data(), evm.size(), h);
            evm.assign(h, h + 32);
            F4 v;
            for (int i = 0; i < 32; ++i) ((uint8_t*)v.l)[i] = h[31 - i];       // the hash as a big-endian integer, mod r
            return fr_to_mont(v);
        }
        const uint8_t z = 0;   // BLAKE2B_PREFIX_CHALLENGE
        st.update(&z, 1);
        uint8_t out[64];
        st.finalize(out);
        return fr_from_uniform(out);
    }
};

}  // namespace host
}  // namespace zk
