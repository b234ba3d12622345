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
#include <string>
#include <vector>
#include <algorithm>
#include <cstdio>
#include <functional>

#include "host_pairing.hpp"

using namespace zk;
using namespace zk::host;

namespace {

// G1 point codecs of halo2curves' SerdeFormat: Processed = 32 B compressed (x little-endian, bit 254 = parity of y,
// bit 255 = identity: host_util.hpp g1_compress), RawBytes* = x || y Montgomery limbs (the in-memory form).
// Reading follows `from_bytes`: an identity flag needs a zero x and a clear parity flag; without it x must be the
// abscissa of a curve point -- 32 zero bytes are refused (x = 0 is on no point of y^2 = x^3 + 3).
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
    F4 xc;
    memcpy(xc.l, in, 32);
    const bool is_inf = (xc.l[3] >> 63) & 1, sign = (xc.l[3] >> 62) & 1;
    xc.l[3] &= ~(3ull << 62);
    if (is_inf) {
        if (sign || (xc.l[0] | xc.l[1] | xc.l[2] | xc.l[3])) return false;
        memset((void*)p, 0, 64);
        return true;
    }
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

// ---- the prover's `Proof` wire object [REF prover/src/proof.rs:25-35, 99-104]: what `dump_as_json` writes to
// full_proof_<name>.json -- {"proof": base64, "instances": base64 of the 32-byte big-endian words, "vk": base64 of
// VerifyingKey::write(Processed), "git_version": string or null}, serde_json's compact form, the field order of the struct;
// base64 = the `base64` crate's `encode` [REF eth-types/src/lib.rs:71-91]: standard alphabet, padded.
static const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
static void b64_encode(const uint8_t* in, size_t n, std::string* out) {
    for (size_t i = 0; i < n; i += 3) {
        const uint32_t a = in[i], b = i + 1 < n ? in[i + 1] : 0u, c = i + 2 < n ? in[i + 2] : 0u, v = (a << 16) | (b << 8) | c;
        out->push_back(B64[v >> 18]);
        out->push_back(B64[(v >> 12) & 63]);
        out->push_back(i + 1 < n ? B64[(v >> 6) & 63] : '=');
        out->push_back(i + 2 < n ? B64[v & 63] : '=');
    }
}
// strict: length a multiple of four, padding only at the end, no stray characters (what the crate's `decode` accepts)
static bool b64_decode(const char* in, size_t n, std::vector<uint8_t>* out) {
    if (n % 4) return false;
    auto val = [](char ch) -> int { const char* p = ch ? strchr(B64, ch) : nullptr; return p ? (int)(p - B64) : -1; };
    for (size_t i = 0; i < n; i += 4) {
        const bool last = i + 4 == n;
        const int pad = last ? (in[i + 3] == '=') + (in[i + 3] == '=' && in[i + 2] == '=') : 0;
        int v[4] = {val(in[i]), val(in[i + 1]), pad >= 2 ? 0 : val(in[i + 2]), pad >= 1 ? 0 : val(in[i + 3])};
        if (v[0] < 0 || v[1] < 0 || v[2] < 0 || v[3] < 0) return false;
        const uint32_t w = ((uint32_t)v[0] << 18) | ((uint32_t)v[1] << 12) | ((uint32_t)v[2] << 6) | (uint32_t)v[3];
        if ((pad >= 1 && (w & 0xFF)) || (pad >= 2 && (w & 0xFFFF))) return false;          // non-canonical trailing bits
        out->push_back((uint8_t)(w >> 16));
        if (pad < 2) out->push_back((uint8_t)(w >> 8));
        if (pad < 1) out->push_back((uint8_t)w);
    }
    return true;
}
int zk_host_proof_json_write(const void* proof, size_t proof_len, const void* instances_be, size_t instances_len, const void* vk, size_t vk_len,
                             const char* git_version, char* out, size_t cap, size_t* len) {
    if (!len || (!proof && proof_len) || (!instances_be && instances_len) || (!vk && vk_len) || instances_len % 32) return ZK_ERR_INVALID_ARG;
    std::string js = "{\"proof\":\"";
    b64_encode((const uint8_t*)proof, proof_len, &js);
    js += "\",\"instances\":\"";
    b64_encode((const uint8_t*)instances_be, instances_len, &js);
    js += "\",\"vk\":\"";
    b64_encode((const uint8_t*)vk, vk_len, &js);
    js += "\",\"git_version\":";
    if (git_version) {
        js += '"';
        for (const char* q = git_version; *q; ++q) {            // serde_json's escapes for what a version string could hold
            const unsigned char ch = (unsigned char)*q;
            if (ch == '"' || ch == '\\') { js += '\\'; js += (char)ch; }
            else if (ch == 0x08) js += "\\b";                    // serde_json's short escapes, then \u00xx for the other controls
            else if (ch == 0x09) js += "\\t";
            else if (ch == 0x0a) js += "\\n";
            else if (ch == 0x0c) js += "\\f";
            else if (ch == 0x0d) js += "\\r";
            else if (ch < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", ch); js += b; }
            else js += (char)ch;
        }
        js += '"';
    } else js += "null";
    js += "}";
    *len = js.size();
    if (!out) return ZK_OK;                                      // size query
    if (cap < js.size()) return ZK_ERR_INVALID_ARG;
    memcpy(out, js.data(), js.size());
    return ZK_OK;
}
// Reads what the function above (or serde_json, compact or pretty) wrote: the four keys in any order.  Other keys are parsed and
// dropped, as serde does for a struct without `deny_unknown_fields` -- the reference flattens `Proof` into ChunkProof / BatchProof
// next to `protocol`, `chunk_info`, `row_usages` [REF prover/src/proof/chunk.rs:10-19], and such an object is accepted here
// (pinned by the reference's own aggregator/data/batch-task.json, tests/test_reference_chunk_proof.py).  A repeated known key is
// refused (serde: "duplicate field").  Each output is optional (NULL buffer = length only); *_len hold the capacities on entry and
// the lengths on return.
int zk_host_proof_json_read(const char* json, size_t json_len, void* proof, size_t* proof_len, void* instances_be, size_t* instances_len, void* vk, size_t* vk_len,
                            char* git_version, size_t git_cap, int* has_git_version) {
    if (!json || !proof_len || !instances_len || !vk_len) return ZK_ERR_INVALID_ARG;
    size_t i = 0;
    auto ws = [&] { while (i < json_len && (json[i] == ' ' || json[i] == '\n' || json[i] == '\r' || json[i] == '\t')) ++i; };
    auto lit = [&](char ch) { ws(); if (i < json_len && json[i] == ch) { ++i; return true; } return false; };
    auto str = [&](std::string* out) -> bool {           // a JSON string with the escapes serde_json writes
        ws();
        if (i >= json_len || json[i] != '"') return false;
        for (++i; i < json_len && json[i] != '"'; ++i) {
            if (json[i] != '\\') { out->push_back(json[i]); continue; }
            if (++i >= json_len) return false;
            switch (json[i]) {
                case '"': case '\\': case '/': out->push_back(json[i]); break;
                case 'n': out->push_back('\n'); break;
                case 't': out->push_back('\t'); break;
                case 'r': out->push_back('\r'); break;
                case 'b': out->push_back('\b'); break;
                case 'f': out->push_back('\f'); break;
                case 'u': {
                    if (i + 4 >= json_len) return false;
                    unsigned v = 0;
                    for (int d = 1; d <= 4; ++d) {
                        const char h = json[i + d];
                        if (!((h >= '0' && h <= '9') || (h >= 'a' && h <= 'f') || (h >= 'A' && h <= 'F'))) return false;
                        v = v * 16 + (unsigned)(h <= '9' ? h - '0' : (h | 0x20) - 'a' + 10);
                    }
                    if (v > 0x7F) return false;          // version strings are ASCII
                    out->push_back((char)v);
                    i += 4;
                    break;
                }
                default: return false;
            }
        }
        if (i >= json_len) return false;
        ++i;
        return true;
    };
    // any JSON value, parsed for well-formedness and dropped (serde's IgnoredAny); depth bounded like serde_json's 128
    std::function<bool(int)> skip = [&](int depth) -> bool {
        if (depth > 128) return false;
        ws();
        if (i >= json_len) return false;
        const char ch = json[i];
        if (ch == '"') {
            for (++i; i < json_len && json[i] != '"'; ++i) {
                if ((unsigned char)json[i] < 0x20) return false;
                if (json[i] != '\\') continue;
                if (++i >= json_len) return false;
                if (json[i] == 'u') {
                    if (i + 4 >= json_len) return false;
                    for (int d = 1; d <= 4; ++d) { const char h = json[i + d]; if (!((h >= '0' && h <= '9') || (h >= 'a' && h <= 'f') || (h >= 'A' && h <= 'F'))) return false; }
                    i += 4;
                } else if (!strchr("\"\\/bfnrt", json[i])) return false;
            }
            if (i >= json_len) return false;
            ++i;
            return true;
        }
        if (ch == '{' || ch == '[') {
            const char close = ch == '{' ? '}' : ']';
            ++i;
            for (bool first = true; !lit(close); first = false) {
                if (!first && !lit(',')) return false;
                if (ch == '{') { ws(); if (i >= json_len || json[i] != '"' || !skip(depth + 1) || !lit(':')) return false; }
                if (!skip(depth + 1)) return false;
            }
            return true;
        }
        for (const char* w : {"true", "false", "null"}) { const size_t l = strlen(w); if (i + l <= json_len && !strncmp(json + i, w, l)) { i += l; return true; } }
        // number: -? (0 | [1-9][0-9]*) (. [0-9]+)? ([eE] [+-]? [0-9]+)?
        size_t j = i;
        auto digits = [&] { const size_t s0 = j; while (j < json_len && json[j] >= '0' && json[j] <= '9') ++j; return j - s0; };
        if (j < json_len && json[j] == '-') ++j;
        if (j < json_len && json[j] == '0') ++j; else if (!digits()) return false;
        if (j < json_len && json[j] == '.') { ++j; if (!digits()) return false; }
        if (j < json_len && (json[j] == 'e' || json[j] == 'E')) { ++j; if (j < json_len && (json[j] == '+' || json[j] == '-')) ++j; if (!digits()) return false; }
        i = j;
        return true;
    };
    std::vector<uint8_t> parts[3];
    bool seen[4] = {false, false, false, false};
    std::string gv;
    bool gv_present = false;
    if (!lit('{')) return ZK_ERR_INVALID_ARG;
    for (bool first = true; !lit('}'); first = false) {
        if (!first && !lit(',')) return ZK_ERR_INVALID_ARG;
        std::string key;
        if (!str(&key) || !lit(':')) return ZK_ERR_INVALID_ARG;
        const int which = key == "proof" ? 0 : key == "instances" ? 1 : key == "vk" ? 2 : key == "git_version" ? 3 : -1;
        if (which < 0) { if (!skip(0)) return ZK_ERR_INVALID_ARG; continue; }
        if (seen[which]) return ZK_ERR_INVALID_ARG;
        seen[which] = true;
        if (which == 3) {
            ws();
            if (i + 4 <= json_len && !strncmp(json + i, "null", 4)) { i += 4; continue; }
            if (!str(&gv)) return ZK_ERR_INVALID_ARG;
            gv_present = true;
            continue;
        }
        std::string b64;
        if (!str(&b64) || !b64_decode(b64.data(), b64.size(), &parts[which])) return ZK_ERR_INVALID_ARG;
    }
    ws();
    if (i != json_len || !seen[0] || !seen[1] || !seen[2]) return ZK_ERR_INVALID_ARG;       // `git_version` may be absent (Option)
    if (parts[1].size() % 32) return ZK_ERR_INVALID_ARG;
    void* outs[3] = {proof, instances_be, vk};
    size_t* lens[3] = {proof_len, instances_len, vk_len};
    for (int p = 0; p < 3; ++p) {
        if (outs[p]) { if (*lens[p] < parts[p].size()) return ZK_ERR_INVALID_ARG; memcpy(outs[p], parts[p].data(), parts[p].size()); }
        *lens[p] = parts[p].size();
    }
    if (has_git_version) *has_git_version = gv_present ? 1 : 0;
    if (git_version && git_cap) { const size_t c = std::min(git_cap - 1, gv.size()); memcpy(git_version, gv.data(), c); git_version[c] = 0; }
    return ZK_OK;
}

// ---- instances as the prover's JSON matrix [REF prover/src/io.rs:28-56]: `serialize_instance` = serde_json of
// Vec<Vec<Vec<u8>>> -- per instance column a list of field elements, each the 32 little-endian bytes of `Fr::to_bytes` written as
// numbers: [[[1,0,...,0],[...]],[...]] in compact form.  (`load_instances` [REF prover/src/io.rs:128-142] reads a list of such
// matrices: wrap / unwrap one more pair of brackets.)
int zk_host_instances_json_write(const void* const* cols_fr_mont, const size_t* lens, size_t ncols, char* out, size_t cap, size_t* len) {
    if (!len || ((!cols_fr_mont || !lens) && ncols)) return ZK_ERR_INVALID_ARG;
    std::string js = "[";
    for (size_t c = 0; c < ncols; ++c) {
        if (!cols_fr_mont[c] && lens[c]) return ZK_ERR_INVALID_ARG;
        js += c ? ",[" : "[";
        for (size_t i = 0; i < lens[c]; ++i) {
            F4 v;
            memcpy(v.l, (const uint8_t*)cols_fr_mont[c] + 32 * i, 32);
            uint8_t le[32];
            fr_to_repr(v, le);
            js += i ? ",[" : "[";
            for (int b = 0; b < 32; ++b) { char t[8]; snprintf(t, sizeof t, b ? ",%u" : "%u", (unsigned)le[b]); js += t; }
            js += "]";
        }
        js += "]";
    }
    js += "]";
    *len = js.size();
    if (!out) return ZK_OK;
    if (cap < js.size()) return ZK_ERR_INVALID_ARG;
    memcpy(out, js.data(), js.size());
    return ZK_OK;
}
// *ncols = number of columns; lens_out[c] (capacity lens_cap) their lengths; fr_mont_out (capacity fr_cap elements) the values of
// all columns one after the other.  NULL outputs = counts only (*total = number of values).  A value >= r, a byte above 255 or
// an element that is not 32 numbers is refused.
int zk_host_instances_json_read(const char* json, size_t json_len, size_t* ncols, size_t* lens_out, size_t lens_cap, void* fr_mont_out, size_t fr_cap, size_t* total) {
    if (!json || !ncols || !total) return ZK_ERR_INVALID_ARG;
    size_t i = 0;
    auto ws = [&] { while (i < json_len && (json[i] == ' ' || json[i] == '\n' || json[i] == '\r' || json[i] == '\t')) ++i; };
    auto lit = [&](char ch) { ws(); if (i < json_len && json[i] == ch) { ++i; return true; } return false; };
    std::vector<size_t> lens;
    std::vector<F4> vals;
    if (!lit('[')) return ZK_ERR_INVALID_ARG;
    for (bool first_c = true; !lit(']'); first_c = false) {
        if (!first_c && !lit(',')) return ZK_ERR_INVALID_ARG;
        if (!lit('[')) return ZK_ERR_INVALID_ARG;
        size_t cnt = 0;
        for (bool first_e = true; !lit(']'); first_e = false) {
            if (!first_e && !lit(',')) return ZK_ERR_INVALID_ARG;
            if (!lit('[')) return ZK_ERR_INVALID_ARG;
            uint8_t le[32];
            for (int b = 0; b < 32; ++b) {
                if (b && !lit(',')) return ZK_ERR_INVALID_ARG;
                ws();
                unsigned v = 0;
                size_t digits = 0;
                const bool lead0 = i < json_len && json[i] == '0';
                while (i < json_len && json[i] >= '0' && json[i] <= '9' && digits < 4) { v = v * 10 + (unsigned)(json[i] - '0'); ++i; ++digits; }
                if (!digits || v > 255 || (lead0 && digits > 1)) return ZK_ERR_INVALID_ARG;      // serde_json refuses leading zeros
                le[b] = (uint8_t)v;
            }
            if (!lit(']')) return ZK_ERR_INVALID_ARG;
            F4 v;
            memcpy(v.l, le, 32);
            if (geq_mod<FrC>(v.l)) return ZK_ERR_INVALID_ARG;            // Fr::from_repr fails
            vals.push_back(fr_to_mont(v));
            ++cnt;
        }
        lens.push_back(cnt);
    }
    ws();
    if (i != json_len) return ZK_ERR_INVALID_ARG;
    *ncols = lens.size();
    *total = vals.size();
    if (lens_out) { if (lens_cap < lens.size()) return ZK_ERR_INVALID_ARG; memcpy(lens_out, lens.data(), lens.size() * sizeof(size_t)); }
    if (fr_mont_out) { if (fr_cap < vals.size()) return ZK_ERR_INVALID_ARG; if (!vals.empty()) memcpy(fr_mont_out, vals.data(), vals.size() * 32); }
    return ZK_OK;
}

}  // extern "C"
