// Host-only entry points around the proving path: the wire formats the
// reference moves keys, instances and accumulators in, and the CPU work the aggregation layers do
// between two GPU proofs.  No context, no device: usable (and tested) without a GPU.
//   * instances as concatenated 32-byte BIG-endian words   [REF prover/src/proof.rs:77-85,126-138]
//   * VerifyingKey::write / read in SerdeFormat::{Processed, RawBytes, RawBytesUnchecked}
//                                                          [REF prover/src/io.rs:97-106]
//   * KZG accumulators: random linear combination over a Poseidon transcript (snark-verifier
//     KzgAs::create_proof without blinding), the decider e(lhs, g2) == e(rhs, s_g2), and the
//     accumulator as 4 x LIMBS = 12 limbs of BITS = 88 bits  [REF aggregator/src/core.rs:48-147],
//                                                          [REF aggregator/src/constants.rs:77-82]
#include "host_pairing.hpp"

using namespace zk;
using namespace zk::host;

namespace {

// G1 point codecs of halo2curves' SerdeFormat: Processed = 32 B compressed (x little-endian, bit 255 =
// parity of y, identity = zeros), RawBytes* = x || y Montgomery limbs (the in-memory form)
size_t point_len(int format) { return format == 0 ? 32 : 64; }
void point_write(const G1Affine& p, int format, uint8_t* out) {
    if (format == 0) g1_compress(p, out);
    else memcpy(out, &p, 64);
}
bool point_read(const uint8_t* in, int format, G1Affine* p) {
    if (format != 0) {
        memcpy((void*)p, in, 64);
        if (format == 2 || p->is_identity()) return true;         // RawBytesUnchecked: taken as is
        F4 x, y, want;
        memcpy(x.l, &p->x, 32);
        memcpy(y.l, &p->y, 32);
        if (geq_mod<FqC>(x.l) || geq_mod<FqC>(y.l)) return false;
        return g1_y_from_x(x, &want) && (memcmp(want.l, y.l, 32) == 0 || memcmp(q_neg(want).l, y.l, 32) == 0);
    }
    bool zero = true;
    for (int i = 0; i < 32; ++i) zero &= in[i] == 0;
    if (zero) { memset((void*)p, 0, 64); return true; }
    F4 xc;
    memcpy(xc.l, in, 32);
    const bool sign = (xc.l[3] >> 63) & 1;
    xc.l[3] &= ~(1ull << 63);
    if (geq_mod<FqC>(xc.l)) return false;
    const F4 x = fmul<FqC>(xc, [] { F4 r2; static const uint64_t R2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}; memcpy(r2.l, R2, 32); return r2; }());
    F4 y;
    if (!g1_y_from_x(x, &y)) return false;
    uint8_t yb[32];
    Fq yq;
    memcpy(&yq, y.l, 32);
    fq_to_repr(yq, yb);
    if ((bool)(yb[0] & 1) != sign) y = q_neg(y);
    memcpy(&p->x, x.l, 32);
    memcpy(&p->y, y.l, 32);
    return true;
}
void put_be32(uint8_t* out, uint32_t v) { out[0] = (uint8_t)(v >> 24); out[1] = (uint8_t)(v >> 16); out[2] = (uint8_t)(v >> 8); out[3] = (uint8_t)v; }
uint32_t get_be32(const uint8_t* in) { return ((uint32_t)in[0] << 24) | ((uint32_t)in[1] << 16) | ((uint32_t)in[2] << 8) | in[3]; }

}  // namespace

extern "C" {

// Fr values (Montgomery) -> n x 32 bytes, canonical, big-endian: `serialize_instance`
int zk_host_instances_encode(const void* fr_mont, size_t n, void* out_be) {
    if ((!fr_mont || !out_be) && n) return ZK_ERR_INVALID_ARG;
    for (size_t i = 0; i < n; ++i) {
        F4 v;
        memcpy(v.l, (const uint8_t*)fr_mont + 32 * i, 32);
        uint8_t le[32];
        fr_to_repr(v, le);
        for (int b = 0; b < 32; ++b) ((uint8_t*)out_be)[32 * i + b] = le[31 - b];
    }
    return ZK_OK;
}
// the inverse (`Proof::instances`); a word >= r is refused (deserialize_fr -> from_repr fails)
int zk_host_instances_decode(const void* in_be, size_t n, void* fr_mont_out) {
    if ((!in_be || !fr_mont_out) && n) return ZK_ERR_INVALID_ARG;
    for (size_t i = 0; i < n; ++i) {
        F4 v;
        for (int b = 0; b < 32; ++b) ((uint8_t*)v.l)[b] = ((const uint8_t*)in_be)[32 * i + 31 - b];
        if (geq_mod<FrC>(v.l)) return ZK_ERR_INVALID_ARG;
        const F4 m = fr_to_mont(v);
        memcpy((uint8_t*)fr_mont_out + 32 * i, m.l, 32);
    }
    return ZK_OK;
}

// G1 points in halo2curves' SerdeFormat (0 Processed, 1 RawBytes, 2 RawBytesUnchecked)
int zk_host_g1_encode(const void* affine64, size_t n, int format, void* out) {
    if (((!affine64 || !out) && n) || format < 0 || format > 2) return ZK_ERR_INVALID_ARG;
    for (size_t i = 0; i < n; ++i) {
        G1Affine p;
        memcpy((void*)&p, (const uint8_t*)affine64 + 64 * i, 64);
        point_write(p, format, (uint8_t*)out + point_len(format) * i);
    }
    return ZK_OK;
}
int zk_host_g1_decode(const void* in, size_t n, int format, void* affine64_out) {
    if (((!in || !affine64_out) && n) || format < 0 || format > 2) return ZK_ERR_INVALID_ARG;
    for (size_t i = 0; i < n; ++i) {
        G1Affine p;
        if (!point_read((const uint8_t*)in + point_len(format) * i, format, &p)) return ZK_ERR_INVALID_ARG;
        memcpy((uint8_t*)affine64_out + 64 * i, &p, 64);
    }
    return ZK_OK;
}

// halo2 VerifyingKey::write: k (u32 BE) | #fixed commitments (u32 BE) | fixed commitments | permutation
// commitments | the selector assignments packed 8 rows per byte (num_selectors x ceil(2^k / 8) bytes,
// as upstream keeps them for the re-compression done by `read`).  The constraint system itself is not
// part of the file: `read` rebuilds it from the circuit type.
int zk_host_vk_write(uint32_t k, const void* fixed_commitments, uint32_t num_fixed, const void* perm_commitments, uint32_t num_perm,
                     const uint8_t* selectors_packed, uint32_t num_selectors, int format, void* out, size_t cap, size_t* len) {
    if (!len || format < 0 || format > 2 || k > 28 || (!fixed_commitments && num_fixed) || (!perm_commitments && num_perm) || (!selectors_packed && num_selectors)) return ZK_ERR_INVALID_ARG;
    const size_t sel_bytes = (size_t)num_selectors * ((((size_t)1 << k) + 7) / 8);
    const size_t need = 8 + point_len(format) * ((size_t)num_fixed + num_perm) + sel_bytes;
    *len = need;
    if (!out || cap < need) return ZK_ERR_INVALID_ARG;
    uint8_t* o = (uint8_t*)out;
    put_be32(o, k);
    put_be32(o + 4, num_fixed);
    o += 8;
    zk_host_g1_encode(fixed_commitments, num_fixed, format, o);
    o += point_len(format) * num_fixed;
    zk_host_g1_encode(perm_commitments, num_perm, format, o);
    o += point_len(format) * num_perm;
    if (sel_bytes) memcpy(o, selectors_packed, sel_bytes);
    return ZK_OK;
}
// the inverse; num_perm / num_selectors come from the circuit's ConstraintSystem as in upstream's `read`
int zk_host_vk_read(const void* in, size_t in_len, int format, uint32_t num_perm, uint32_t num_selectors, uint32_t* k, uint32_t* num_fixed,
                    void* fixed_commitments, size_t fixed_cap, void* perm_commitments, uint8_t* selectors_packed) {
    if (!in || !k || !num_fixed || format < 0 || format > 2 || in_len < 8) return ZK_ERR_INVALID_ARG;
    const uint8_t* p = (const uint8_t*)in;
    *k = get_be32(p);
    *num_fixed = get_be32(p + 4);
    if (*k > 28) return ZK_ERR_INVALID_ARG;
    const size_t sel_bytes = (size_t)num_selectors * ((((size_t)1 << *k) + 7) / 8);
    if (in_len != 8 + point_len(format) * ((size_t)*num_fixed + num_perm) + sel_bytes) return ZK_ERR_INVALID_ARG;
    if (*num_fixed > fixed_cap || (!fixed_commitments && *num_fixed) || (!perm_commitments && num_perm)) return ZK_ERR_INVALID_ARG;
    p += 8;
    if (int rc = zk_host_g1_decode(p, *num_fixed, format, fixed_commitments)) return rc;
    p += point_len(format) * *num_fixed;
    if (int rc = zk_host_g1_decode(p, num_perm, format, perm_commitments)) return rc;
    p += point_len(format) * num_perm;
    if (sel_bytes && selectors_packed) memcpy(selectors_packed, p, sel_bytes);
    return ZK_OK;
}

// prod_i e(P_i, Q_i) == 1 ?  P: n x 64 B G1Affine, Q: n x 128 B G2Affine (x.c0, x.c1, y.c0, y.c1), Montgomery limbs.
// *ok receives the verdict.
int zk_host_pairing_check(const void* g1_points, const void* g2_points, size_t n, int* ok) {
    if (!ok || ((!g1_points || !g2_points) && n)) return ZK_ERR_INVALID_ARG;
    std::vector<G1Affine> P(n);
    std::vector<G2Affine> Q(n);
    if (n) { memcpy((void*)P.data(), g1_points, 64 * n); memcpy((void*)Q.data(), g2_points, 128 * n); }
    *ok = pairing_check(P.data(), Q.data(), n) ? 1 : 0;
    return ZK_OK;
}
// the KZG decider of the aggregation layers: e(lhs, g2) == e(rhs, s_g2)
int zk_host_accumulator_check(const void* lhs, const void* rhs, const void* g2, const void* s_g2, int* ok) {
    if (!lhs || !rhs || !g2 || !s_g2 || !ok) return ZK_ERR_INVALID_ARG;
    G1Affine P[2];
    G2Affine Q[2];
    memcpy((void*)&P[0], lhs, 64);
    memcpy((void*)&P[1], rhs, 64);
    memcpy((void*)&Q[0], g2, 128);
    memcpy((void*)&Q[1], s_g2, 128);
    F4 y;
    memcpy(y.l, &P[1].y, 32);
    y = q_neg(y);                                              // e(lhs, g2) * e(-rhs, s_g2) == 1
    if (!P[1].is_identity()) memcpy(&P[1].y, y.l, 32);
    *ok = pairing_check(P, Q, 2) ? 1 : 0;
    return ZK_OK;
}
// snark-verifier KzgAs::create_proof (no blinding): absorb every (lhs_i, rhs_i) into a Poseidon transcript,
// r = squeeze, (lhs, rhs) = (sum r^i lhs_i, sum r^i rhs_i).  lhs_in / rhs_in: n x 64 B; r_out (nullable): Fr.
int zk_host_accumulate(const void* lhs_in, const void* rhs_in, size_t n, void* lhs_out, void* rhs_out, void* r_out) {
    if (!lhs_in || !rhs_in || !lhs_out || !rhs_out || n == 0) return ZK_ERR_INVALID_ARG;
    std::vector<G1Affine> L(n), Rr(n);
    memcpy((void*)L.data(), lhs_in, 64 * n);
    memcpy((void*)Rr.data(), rhs_in, 64 * n);
    Transcript tr;
    tr.reset(ZK_TRANSCRIPT_POSEIDON);
    for (size_t i = 0; i < n; ++i) { tr.common_point(L[i]); tr.common_point(Rr[i]); }
    if (tr.err) return ZK_ERR_INVALID_ARG;                     // an identity accumulator cannot be absorbed (upstream errors too)
    const F4 r = tr.squeeze();
    if (r_out) memcpy(r_out, r.l, 32);
    PXyzz accL, accR;
    memset(&accL, 0, sizeof accL);
    memset(&accR, 0, sizeof accR);
    F4 pw = fr_one();
    for (size_t i = 0; i < n; ++i) {
        const F4 k = fr_canon(pw);
        accL = padd(accL, g1_mul_canon(L[i], k));
        accR = padd(accR, g1_mul_canon(Rr[i], k));
        pw = fr_mul(pw, r);
    }
    pto_affine(accL, (G1Affine*)lhs_out);
    pto_affine(accR, (G1Affine*)rhs_out);
    return ZK_OK;
}
// [lhs.x, lhs.y, rhs.x, rhs.y] as 3 limbs of 88 bits each, least significant first, every limb an Fr
// (Montgomery): the 12 instance cells an aggregation circuit exposes first
int zk_host_accumulator_limbs(const void* lhs, const void* rhs, void* out12_fr) {
    if (!lhs || !rhs || !out12_fr) return ZK_ERR_INVALID_ARG;
    G1Affine p[2];
    memcpy((void*)&p[0], lhs, 64);
    memcpy((void*)&p[1], rhs, 64);
    int o = 0;
    for (int j = 0; j < 2; ++j) {
        for (const Fq* coord : {&p[j].x, &p[j].y}) {
            uint8_t b[32];
            fq_to_repr(*coord, b);                                      // canonical, little-endian
            for (int limb = 0; limb < 3; ++limb) {
                F4 v = fr_zero();
                const int nbytes = limb < 2 ? 11 : 10;                  // 88 + 88 + 80 bits
                memcpy(v.l, b + 11 * limb, nbytes);
                const F4 m = fr_to_mont(v);
                memcpy((uint8_t*)out12_fr + 32 * o++, m.l, 32);
            }
        }
    }
    return ZK_OK;
}

}  // extern "C"
