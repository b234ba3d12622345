// Host orchestration of a full PLONKish/KZG proof on top of the device kernels: the surface of
// halo2_proofs::plonk::{keygen_pk, create_proof} with the GWC (poly::kzg::multiopen::ProverGWC,
// what BASELINE.json's north_star names) or SHPLONK (ProverSHPLONK, what the reference's call
// sites instantiate) multi-open.  External crate;
// reference call sites [REF circuit-benchmarks/src/super_circuit.rs:109-132],
// [REF prover/src/common/prover/utils.rs:31,55].
//
// What runs where:
//   device : every commitment (MSM), every NTT / coset NTT, expression evaluation over Lagrange
//            rows and over the cosets of the extended domain (quotient.hip), lookup multiplicities
//            (lookup.hip), batch inversion, grand product / grand sum scans, the blinding
//            polynomial (ChaCha20 counter mode), polynomial evaluation, Kate division, linear
//            combinations
//   host   : transcript (built-in Blake2b or the caller's object through a callback table), the
//            blinding-row RNG (XorShift), program assembly, a handful of scalar field operations
//            per challenge; for multi-GPU sessions the all-gather callbacks
//
// Protocol (mv-lookup/logUp lookups as in the Scroll halo2 fork, SURVEY note L):
//   vk_repr, instances | per phase: advice commitments, the phase's challenges | theta |
//   m commitments | beta, gamma | permutation Z commitments | lookup phi commitments |
//   random poly | y | h pieces | x | evaluations | GWC: v, witnesses  or  SHPLONK: y, v, h, u, pi.
// The matching verifier is oracle/plonk_verifier.py; oracle/plonk_prover.py restates this file over
// Python integers and the tests require both to produce the same proof bytes.
//
// The circuit arrives as a flat "pk blob" (zkevm-circuits_amd/plonk.py serialises it, the Rust shim
// fills it from halo2's ConstraintSystem; layout in INTEGRATION.md, SURVEY 8f-1 export format):
// header, phases, the advice / fixed / instance query lists in halo2's registration order (the order
// of the evaluations in the proof), permutation columns, constants, gate programs, lookup arguments
// (one table tuple and one or more input tuples each, as chunk_lookups() leaves them), fixed columns
// and sigma columns in Lagrange form.
#include <algorithm>
#include <array>
#include <unordered_map>
#include <unordered_set>
#include <memory>
#include <string>

#include <chrono>
#include <cstdlib>
#include <random>
#include <tuple>
#include <functional>
#include "ctx.hpp"
#include "host_hash.hpp"

using namespace zk;
using zk::host::F4;

extern "C" int zk_quotient_eval(zk_ctx*, const uint32_t*, uint32_t, const void* const*, uint32_t, const void*, uint32_t, uint32_t, uint32_t, int, void*);
extern "C" int zk_fr_powers(zk_ctx*, const void*, const void*, void*, size_t);
extern "C" int zk_ntt(zk_ctx*, void*, uint32_t, int);
extern "C" int zk_fr_batch_invert(zk_ctx*, void*, size_t);
extern "C" int zk_fr_random(zk_ctx*, const uint8_t*, uint64_t, uint64_t, void*, size_t);
extern "C" int zk_lookup_multiplicities(zk_ctx*, const void*, const void*, size_t, void*, size_t, uint64_t*);
extern "C" int zk_coeff_to_coset(zk_ctx*, const void*, uint32_t, const void*, void*);
extern "C" int zk_coeff_to_coset_batch(zk_ctx*, const void* const*, uint32_t, const void*, void* const*, size_t);
extern "C" int zk_fr_scatter_scaled(zk_ctx*, const void*, size_t, const void*, void*, size_t, size_t);
extern "C" int zk_msm_g1(zk_ctx*, const void*, const void*, size_t, void*);
extern "C" int zk_g1_sum_host(const void*, size_t, void*);
extern "C" int zk_poly_eval_batch(zk_ctx*, const void* const*, size_t, size_t, const void*, void*);
extern "C" int zk_poly_eval_pairs(zk_ctx*, const void* const*, const uint32_t*, size_t, const void*, size_t, size_t, void*);

namespace {

enum ColType : uint32_t { CT_FIXED = 0, CT_ADVICE = 1, CT_INSTANCE = 2, CT_SPECIAL = 3, CT_PERM_Z = 4, CT_SIGMA = 5, CT_LK_M = 6, CT_LK_PHI = 7, CT_RANDOM = 8, CT_H = 9, CT_SPLIT_R = 10 /* remainder polynomials of the additive split (zk_proof_finish) */ };
enum Special : uint32_t { SP_X = 0, SP_L0 = 1, SP_LLAST = 2, SP_LACTIVE = 3 };
enum QOp : uint32_t { Q_END = 0, Q_PUSH_COL = 1, Q_PUSH_CONST = 2, Q_ADD = 3, Q_SUB = 4, Q_MUL = 5, Q_NEG = 6, Q_SQUARE = 7, Q_DOUBLE = 8, Q_FOLD = 9, Q_MUL_CONST = 10, Q_ADD_CONST = 11, Q_TEE_TMP = 12, Q_PUSH_TMP = 13 };
// abstract constant operands: user constants are [0, num_consts); challenges live above
constexpr uint32_t C_THETA = 0xFFFF0000u, C_BETA = 0xFFFF0001u, C_GAMMA = 0xFFFF0002u, C_Y = 0xFFFF0003u, C_ONE = 0xFFFF0004u, C_ZERO = 0xFFFF0005u, C_DELTA0 = 0xFFFE0000u,   // C_DELTA0 + j = beta * delta^j
                   C_CHAL0 = 0xFFFD0000u,                                                                                      // C_CHAL0 + i = user challenge i
                   C_YPOW0 = 0xFFFC0000u;                                                                                      // C_YPOW0 + g = y^g (folding constraints that are g positions apart)

inline uint32_t colref(uint32_t type, uint32_t idx) { return (type << 24) | idx; }

struct Instr { uint32_t op, a, b; };
typedef std::vector<Instr> Prog;

// Inside a PoolScope (the proof-session entry points) DevBuf blocks come from and return to the
// context's block pool; outside (key generation: buffers that live as long as the key) they are
// plain hipMalloc / hipFree.  A session must not outlive its context.
static thread_local zk_ctx* tl_pool_ctx = nullptr;
struct PoolScope {
    zk_ctx* prev;
    explicit PoolScope(zk_ctx* c) : prev(tl_pool_ctx) { tl_pool_ctx = c; }
    ~PoolScope() { tl_pool_ctx = prev; }
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    zk_ctx* owner = nullptr;
    bool borrowed = false;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), owner(o.owner), borrowed(o.borrowed) { o.p = nullptr; o.borrowed = false; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; bytes = o.bytes; owner = o.owner; borrowed = o.borrowed; o.p = nullptr; o.borrowed = false; } return *this; }
    ~DevBuf() { release(); }
    void release() {
        if (!p) return;
        if (borrowed) { p = nullptr; borrowed = false; return; }
        if (owner) owner->pool_put(p, bytes); else (void)hipFree(p);
        p = nullptr;
    }
    // a view of somebody else's block (never released through this object)
    void borrow(void* q) { release(); p = q; borrowed = true; }
    bool alloc(size_t n) {
        release();
        bytes = n;
        owner = tl_pool_ctx;
        if (owner) { p = owner->pool_get(n); return p != nullptr; }
        return hipMalloc(&p, n ? n : 1) == hipSuccess;
    }
    // plain hipMalloc / hipFree even inside a PoolScope (buffers that outlive the session, e.g. the key's coset cache)
    bool alloc_unpooled(size_t n) {
        release();
        bytes = n;
        owner = nullptr;
        return hipMalloc(&p, n ? n : 1) == hipSuccess;
    }
    Fr* fr() const { return (Fr*)p; }
};

// page-locked host staging (blinding rows): an asynchronous copy from pageable memory blocks the
// host until the stream reaches it, which would serialise the upload pipeline with the enqueueing thread
struct PinnedBuf {
    void* p = nullptr;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    bool alloc(size_t bytes) { return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess; }
};

struct Query { uint32_t type, idx; int32_t rot; };

// ZK_PROVER_TRACE=1: wall-clock per prover stage on stderr (device drained at every mark)
struct StageTrace {
    zk_ctx* ctx; bool on; std::chrono::steady_clock::time_point t0;
    explicit StageTrace(zk_ctx* c) : ctx(c), on(getenv("ZK_PROVER_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what) {
        if (!on) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[zk prover] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

}  // namespace

struct QuotientPlan;
struct zk_pk {
    uint32_t k = 0, bf = 0, d = 0, ext_k = 0, F = 0, A = 0, I = 0, P = 0, L = 0;
    uint32_t chunk = 0, C = 0, u = 0;   // permutation chunk size, #chunks, last usable row index
    std::vector<std::pair<uint32_t, uint32_t>> perm_cols;
    std::vector<F4> consts;
    std::vector<Prog> gates;
    struct Lookup { std::vector<Prog> tables; std::vector<std::vector<Prog>> inputs; };      // mv_lookup::Argument: table_expressions, inputs_expressions
    std::vector<Lookup> lookups;
    std::vector<Query> adv_q, fix_q;         // evaluation queries, in proof order
    uint32_t num_phases = 1;
    std::vector<uint32_t> adv_phase;         // phase of every advice column (halo2 FirstPhase/SecondPhase/...)
    std::vector<uint32_t> chal_phase;        // challenge i becomes available after this phase
    // device-resident key material
    std::vector<DevBuf> fixed_lag, fixed_coeff, sigma_lag, sigma_coeff;
    DevBuf omega_lag, l0_lag, llast_lag, lactive_lag, l0_coeff, llast_coeff, lactive_coeff;
    // cosets of the key's own columns (fixed, sigma, l0 / l_last / l_active, X), filled by the first
    // proof and reused by later ones when they fit the budget (ZK_PK_COSET_CACHE_GB, default 96):
    // part_cache[r][column reference].  Plain device allocations owned by the key (a slot appears
    // only once it is filled).  A key serves one proving thread at a time, as its context does.
    mutable std::vector<std::unordered_map<uint32_t, DevBuf>> part_cache;
    mutable int part_cache_state = -1;       // -1 undecided, 0 off, 1 on, 2 frozen (slots that exist are used, no new ones: an allocation failed)
    mutable size_t part_cache_bytes = 0;     // what this key's slots hold of the context's shared budget (zk_ctx::coset_cache_bytes)
    std::vector<G1Affine> fixed_com, sigma_com;
    F4 vk_repr;                              // vk.transcript_repr: the default, or what zk_pk_set_transcript_repr installed
    std::vector<Query> inst_q;               // instance queries (verifier side; carried for the vk)
    const zk_srs* srs = nullptr;
    mutable std::shared_ptr<const QuotientPlan> qplan;      // the quotient's plan (degree classes, class programs), made on first use
};

struct zk_proof {
    const zk_pk* pk;
    host::XorShiftRng rng;
    host::Transcript tr;
    std::vector<DevBuf> inst_lag, inst_coeff, adv_lag, adv_coeff;
    // Cosets of advice columns computed AHEAD, during the advice phases (the uploads are PCIe-bound and leave the device idle for
    // part of every column; which coset reads which column is a property of the key, advice_coset_plan): adv_coset[r][column],
    // empty where nothing was precomputed.  pre_mask[column] = cosets to precompute (bit r), chosen once per session.
    std::vector<std::vector<DevBuf>> adv_coset;
    std::vector<uint32_t> pre_mask;
    bool pre_planned = false;
    uint32_t phase = 0;
    int multiopen = ZK_MULTIOPEN_GWC;
    int vanishing_random = ZK_VANISHING_ONE;
    bool advice_on_device = false;            // the witness came as device buffers (zk_proof_advice_phase_dev)
    bool advice_in_place = false;             // the Lagrange forms of the advice columns are the caller's device buffers (zk_proof_advice_phase_dev, IN_PLACE)
    bool lag_partial = false;                 // sharded session: some advice columns of other ranks arrived in coefficient form only (nothing on this side of the boundary reads their Lagrange form)
    // multi-GPU sharding (one process per GPU, every rank runs the same session on the same inputs):
    // commitments and quotient cosets are split over the ranks, results exchanged through `gather`
    std::vector<F4> absorbed;                // what zk_proof_begin fed the transcript (replayed into an external one)
    zk_transcript_vtable ext_vt{};           // copy of the caller's vtable when an external transcript is set
    uint32_t rank = 0, world = 1;
    zk_allgather_fn gather = nullptr;
    void* gather_user = nullptr;
    bool use_comm = false;                   // exchanges go through the context's RCCL communicator (zk_proof_set_sharding_comm)
    zk_allgather_fn gather_dev = nullptr;    // optional: all-gather of DEVICE buffers (RCCL over xGMI) for the advice columns
    void* gather_dev_user = nullptr;
    std::vector<F4> challenges;
    zk_proof(const zk_pk* k, const uint8_t* seed) : pk(k), rng(seed), inst_lag(k->I), inst_coeff(k->I), adv_lag(k->A), adv_coeff(k->A), challenges(k->chal_phase.size(), host::fr_zero()) {}
};

namespace {

#define PK_TRY(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)

struct Reader {
    const uint8_t* p; size_t left; bool ok = true;
    uint32_t u32() { if (left < 4) { ok = false; return 0; } uint32_t v; memcpy(&v, p, 4); p += 4; left -= 4; return v; }
    const uint8_t* bytes(size_t n) { if (left < n) { ok = false; return nullptr; } const uint8_t* r = p; p += n; left -= n; return r; }
    // a count of records of `each` bytes that the rest of the blob can actually hold (a truncated or
    // hostile header must not drive allocations)
    uint32_t count(size_t each) { const uint32_t c = u32(); if (ok && (size_t)c * each > left) ok = false; return ok ? c : 0; }
    Prog prog() { Prog g; const uint32_t len = count(12); g.reserve(len); for (uint32_t i = 0; i < len && ok; ++i) { Instr in; in.op = u32(); in.a = u32(); in.b = u32(); g.push_back(in); } return g; }
    void queries(std::vector<Query>* out, uint32_t type, uint32_t ncols) {
        const uint32_t cnt = count(8);
        for (uint32_t i = 0; i < cnt && ok; ++i) { const uint32_t c = u32(); const int32_t rot = (int32_t)u32(); if (c >= ncols) ok = false; out->push_back(Query{type, c, rot}); }
    }
};

// Degree of a postfix program as halo2's Expression::degree computes it (columns 1, constants and
// challenges 0, sums the maximum, products the sum); -1 on a malformed program.
int program_degree(const Prog& g, std::vector<int>* tmp_degree) {
    std::vector<int> st;
    for (const Instr& in : g) {
        switch (in.op) {
            case Q_PUSH_COL: st.push_back(1); break;
            case Q_PUSH_CONST: st.push_back(0); break;
            case Q_ADD: case Q_SUB: if (st.size() < 2) return -1; { const int b_ = st.back(); st.pop_back(); st.back() = std::max(st.back(), b_); } break;
            case Q_MUL: if (st.size() < 2) return -1; { const int b_ = st.back(); st.pop_back(); st.back() += b_; } break;
            case Q_NEG: case Q_DOUBLE: case Q_ADD_CONST: case Q_MUL_CONST: if (st.empty()) return -1; break;
            case Q_SQUARE: if (st.empty()) return -1; st.back() *= 2; break;
            case Q_TEE_TMP: if (st.empty()) return -1; if (in.a >= tmp_degree->size()) tmp_degree->resize(in.a + 1, 0); (*tmp_degree)[in.a] = st.back(); break;
            case Q_PUSH_TMP: if (in.a >= tmp_degree->size()) return -1; st.push_back((*tmp_degree)[in.a]); break;
            default: return -1;
        }
    }
    return st.size() == 1 ? st[0] : -1;
}

int commit_lagrange(zk_ctx* ctx, const zk_srs* srs, const Fr* d_vals, size_t n, G1Affine* out) { return zk_commit(ctx, srs, 1, d_vals, n, out); }
int commit_coeff(zk_ctx* ctx, const zk_srs* srs, const Fr* d_vals, size_t n, G1Affine* out) { return zk_commit(ctx, srs, 0, d_vals, n, out); }

// Lagrange values -> coefficients (EvaluationDomain::lagrange_to_coeff)
int to_coeff(zk_ctx* ctx, const zk_pk* pk, const DevBuf& lag, DevBuf* coeff) {
    const size_t n = (size_t)1 << pk->k;
    if (!coeff->alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
    const Fr omega_inv = fr_inv_host(fr_root_of_unity(pk->k)), ninv = fr_inv_host(fr_from_u64(1ull << pk->k));
    return ntt_run(ctx, coeff->fr(), pk->k, omega_inv, &ninv, nullptr, nullptr, lag.fr());
}

// to_coeff on the auxiliary stream (the caller orders that stream after the column's upload and the
// main stream after it again): the advice phase keeps its main stream for the commitment pipeline
int to_coeff_aux(zk_ctx* ctx, const zk_pk* pk, const DevBuf& lag, DevBuf* coeff) {
    hipStream_t main_stream = ctx->stream;
    ctx->stream = ctx->stream_aux;
    const int rc = to_coeff(ctx, pk, lag, coeff);
    ctx->stream = main_stream;
    return rc;
}

// ------------------------------------------------------------------------------------------------
// Program concretisation: abstract column references -> indices into a flat pointer table,
// abstract constants -> indices into a flat constant table.
struct Concrete {
    std::vector<uint32_t> words;
    std::vector<const void*> cols;
    std::vector<F4> consts;
    std::unordered_map<uint32_t, uint32_t> col_index, const_index;
};

struct Env {   // where an abstract column lives in the form being read (Lagrange values or coefficients)
    const zk_pk* pk;
    const std::unordered_map<uint32_t, const void*>* part;   // non-null: one coset of the extended domain, by column reference
    const std::vector<DevBuf>* advice;
    const std::vector<DevBuf>* instance;
    const std::vector<DevBuf>* perm_z;
    const std::vector<DevBuf>* lk_m;
    const std::vector<DevBuf>* lk_phi;
    F4 theta, beta, gamma, y;
    std::vector<F4> beta_delta;   // beta * delta^j
    std::vector<F4> challenges;   // user challenges (halo2 `Challenge`), by index
    std::vector<F4> ypow;         // y^g, g < size (optional: the weights of a grouped class program, assemble_grouped)
};

const void* resolve_col(const Env& e, uint32_t ref) {
    const uint32_t type = ref >> 24, idx = ref & 0xFFFFFF;
    const zk_pk* pk = e.pk;
    if (e.part) { auto it = e.part->find(ref); return it == e.part->end() ? nullptr : it->second; }
    switch (type) {
        case CT_FIXED: return idx < pk->F ? pk->fixed_lag[idx].p : nullptr;
        case CT_ADVICE: return e.advice && idx < e.advice->size() ? (*e.advice)[idx].p : nullptr;
        case CT_INSTANCE: return e.instance && idx < e.instance->size() ? (*e.instance)[idx].p : nullptr;
        case CT_SPECIAL:
            if (idx == SP_X) return pk->omega_lag.p;
            if (idx == SP_L0) return pk->l0_lag.p;
            if (idx == SP_LLAST) return pk->llast_lag.p;
            if (idx == SP_LACTIVE) return pk->lactive_lag.p;
            return nullptr;
        case CT_PERM_Z: return e.perm_z && idx < e.perm_z->size() ? (*e.perm_z)[idx].p : nullptr;
        case CT_SIGMA: return idx < pk->P ? pk->sigma_lag[idx].p : nullptr;
        case CT_LK_M: return e.lk_m && idx < e.lk_m->size() ? (*e.lk_m)[idx].p : nullptr;
        case CT_LK_PHI: return e.lk_phi && idx < e.lk_phi->size() ? (*e.lk_phi)[idx].p : nullptr;
        default: return nullptr;
    }
}
bool resolve_const(const Env& e, uint32_t ref, F4* out) {
    if (ref < e.pk->consts.size()) { *out = e.pk->consts[ref]; return true; }
    switch (ref) {
        case C_THETA: *out = e.theta; return true;
        case C_BETA: *out = e.beta; return true;
        case C_GAMMA: *out = e.gamma; return true;
        case C_Y: *out = e.y; return true;
        case C_ONE: *out = host::fr_one(); return true;
        case C_ZERO: *out = host::fr_zero(); return true;
        default: break;
    }
    if (ref >= C_DELTA0 && ref - C_DELTA0 < e.beta_delta.size()) { *out = e.beta_delta[ref - C_DELTA0]; return true; }
    if (ref >= C_CHAL0 && ref - C_CHAL0 < e.challenges.size()) { *out = e.challenges[ref - C_CHAL0]; return true; }
    if (ref >= C_YPOW0 && ref < C_CHAL0) { *out = ref - C_YPOW0 < e.ypow.size() ? e.ypow[ref - C_YPOW0] : host::fr_pow(e.y, ref - C_YPOW0); return true; }
    return false;
}
int concretise(zk_ctx* ctx, const Env& e, const Prog& g, Concrete* c) {
    for (const Instr& in : g) {
        uint32_t a = in.a;
        if (in.op == Q_PUSH_COL) {
            auto it = c->col_index.find(in.a);
            if (it == c->col_index.end()) {
                const void* p = resolve_col(e, in.a);
                if (!p) return ctx->fail(ZK_ERR_INVALID_ARG, "prover: unresolved column reference 0x%08x", in.a);
                a = (uint32_t)c->cols.size();
                c->cols.push_back(p);
                c->col_index[in.a] = a;
            } else a = it->second;
        } else if (in.op == Q_PUSH_CONST || in.op == Q_FOLD || in.op == Q_MUL_CONST || in.op == Q_ADD_CONST) {
            auto it = c->const_index.find(in.a);
            if (it == c->const_index.end()) {
                F4 v;
                if (!resolve_const(e, in.a, &v)) return ctx->fail(ZK_ERR_INVALID_ARG, "prover: unresolved constant reference 0x%08x", in.a);
                a = (uint32_t)c->consts.size();
                c->consts.push_back(v);
                c->const_index[in.a] = a;
            } else a = it->second;
        }
        c->words.push_back(in.op); c->words.push_back(a); c->words.push_back(in.b);
    }
    return ZK_OK;
}
// Every form the prover evaluates over has n = 2^k rows (Lagrange values, or one coset of the
// extended domain), so a rotation is a plain index shift modulo n.
int run_program(zk_ctx* ctx, const Env& e, const Prog& g, void* d_out) {
    Concrete c;
    PK_TRY(concretise(ctx, e, g, &c));
    {   // measurement knob (results are WRONG): every operand read from ONE column -- the same instruction stream with its loads served by the caches
        static const int alias = getenv("ZK_QUOTIENT_ALIAS") ? atoi(getenv("ZK_QUOTIENT_ALIAS")) : 0;      // N: the class programs read N distinct columns in all
        if (alias > 0 && ctx->prof_tag && !strcmp(ctx->prof_tag, "quotient_coset")) for (size_t i = 0; i < c.cols.size(); ++i) c.cols[i] = c.cols[i % (size_t)alias];
    }
    return zk_quotient_eval(ctx, c.words.data(), (uint32_t)(c.words.size() / 3), c.cols.data(), (uint32_t)c.cols.size(),
                            c.consts.empty() ? nullptr : c.consts.data(), (uint32_t)c.consts.size(), e.pk->k, e.pk->k, 0, d_out);
}
// coefficient form of an abstract column (nullptr for X, which has no stored polynomial)
const Fr* coeff_of(const zk_pk* pk, uint32_t ref, const std::vector<DevBuf>& adv, const std::vector<DevBuf>& inst, const std::vector<DevBuf>& pz,
                   const std::vector<DevBuf>& lkm, const std::vector<DevBuf>& lkphi) {
    const uint32_t type = ref >> 24, idx = ref & 0xFFFFFF;
    auto at = [&](const std::vector<DevBuf>& v) -> const Fr* { return idx < v.size() ? v[idx].fr() : nullptr; };
    switch (type) {
        case CT_FIXED: return at(pk->fixed_coeff);
        case CT_ADVICE: return at(adv);
        case CT_INSTANCE: return at(inst);
        case CT_SPECIAL: return idx == SP_L0 ? pk->l0_coeff.fr() : idx == SP_LLAST ? pk->llast_coeff.fr() : idx == SP_LACTIVE ? pk->lactive_coeff.fr() : nullptr;
        case CT_PERM_Z: return at(pz);
        case CT_SIGMA: return at(pk->sigma_coeff);
        case CT_LK_M: return at(lkm);
        case CT_LK_PHI: return at(lkphi);
        default: return nullptr;
    }
}

// ---- small program builder ----------------------------------------------------------------------
struct PB {
    Prog g;
    PB& col(uint32_t type, uint32_t idx, int32_t rot = 0) { g.push_back({Q_PUSH_COL, colref(type, idx), (uint32_t)rot}); return *this; }
    PB& cst(uint32_t ref) { g.push_back({Q_PUSH_CONST, ref, 0}); return *this; }
    PB& op(uint32_t o) { g.push_back({o, 0, 0}); return *this; }
    PB& mulc(uint32_t ref) { g.push_back({Q_MUL_CONST, ref, 0}); return *this; }
    PB& addc(uint32_t ref) { g.push_back({Q_ADD_CONST, ref, 0}); return *this; }
    PB& fold(uint32_t ref) { g.push_back({Q_FOLD, ref, 0}); return *this; }
    PB& append(const Prog& o) { g.insert(g.end(), o.begin(), o.end()); return *this; }
};
// theta-compression of a list of expressions: ((e0 * theta + e1) * theta + e2) ...
// An expression of the form F * X or X * F with F a single column read: X (postfix) and F; false for any other shape.
// (The inputs of a zkEVM lookup are `q_enable * value`, all of a tuple under the same q_enable.)
static bool split_leaf_factor(const Prog& g, bool factor_first, Prog* rest, Instr* factor) {
    if (g.size() < 3 || g.back().op != Q_MUL) return false;
    if (factor_first) {
        if (g[0].op != Q_PUSH_COL) return false;
        *factor = g[0];
        rest->assign(g.begin() + 1, g.end() - 1);
    } else {
        if (g[g.size() - 2].op != Q_PUSH_COL) return false;
        *factor = g[g.size() - 2];
        rest->assign(g.begin(), g.end() - 2);
    }
    // `rest` must be ONE complete expression (the other operand of the product at the root)
    int sp = 0;
    for (const Instr& in : *rest) {
        switch (in.op) {
            case Q_PUSH_COL: case Q_PUSH_CONST: case Q_PUSH_TMP: ++sp; break;
            case Q_ADD: case Q_SUB: case Q_MUL: if (sp < 2) return false; --sp; break;
            case Q_TEE_TMP: return false;
            default: if (sp < 1) return false; break;
        }
    }
    return sp == 1;
}
void push_compressed(PB& b, const std::vector<Prog>& exprs) {
    // sum_i theta^(N-1-i) (F x_i) = F * sum_i theta^(N-1-i) x_i: one product by F instead of N (exact: same polynomial)
    if (exprs.size() >= 2) {
        for (int first = 1; first >= 0; --first) {
            std::vector<Prog> rest(exprs.size());
            Instr f0{0, 0, 0};
            bool all = true;
            for (size_t i = 0; i < exprs.size() && all; ++i) {
                Instr f{0, 0, 0};
                all = split_leaf_factor(exprs[i], first != 0, &rest[i], &f) && (i == 0 || (f.a == f0.a && f.b == f0.b));
                if (i == 0) f0 = f;
            }
            if (!all) continue;
            for (size_t i = 0; i < rest.size(); ++i) {
                if (i) b.mulc(C_THETA);
                b.append(rest[i]);
                if (i) b.op(Q_ADD);
            }
            b.g.push_back(f0);
            b.op(Q_MUL);
            return;
        }
    }
    for (size_t i = 0; i < exprs.size(); ++i) {
        if (i) b.mulc(C_THETA);
        b.append(exprs[i]);
        if (i) b.op(Q_ADD);
    }
}
void push_perm_col(PB& b, const std::pair<uint32_t, uint32_t>& c) { b.col(c.first, c.second, 0); }

// usable-row masks in Lagrange form
// `count` commitments over one basis, split over the ranks of a sharded session: rank r commits
// columns i = r, r + world, ... (pipelined batch) and the 64-byte points are all-gathered, so every
// rank ends up with all of them in order and the transcripts stay identical.
int sharded_commit(zk_ctx* ctx, const zk_proof* pr, const zk_srs* srs, int basis, const void* const* ptrs, size_t count, size_t n, G1Affine* out, uint8_t kind = 0 /* zk_commit_batch_hint: 0 dense, 1 small values, 2 runs of equal values, 3 sums with mostly equal increments */) {
    const std::vector<uint8_t> hint(count, kind);
    if (pr->world <= 1 || !pr->gather) return commit_batch_staged(ctx, srs, basis, ptrs, count, n, out, nullptr, nullptr, hint.data());
    if (count < pr->world && n >= ((size_t)pr->world << 10)) {
        // Fewer columns than ranks (aggregation layers: ~10 columns at k = 22..25, h pieces, the closing
        // commitments): shard every MSM by POINTS instead -- rank r takes the r-th slice of the bases of
        // each column, the 64-byte partial results are all-gathered (RCCL has no elliptic-curve reduction)
        // and summed on the host.
        const size_t base = n / pr->world, rem = n % pr->world;
        const size_t lo = pr->rank * base + std::min<size_t>(pr->rank, rem), len = base + (pr->rank < rem ? 1 : 0);
        const G1Affine* bases = basis ? srs->g_lagrange : srs->g;
        std::vector<G1Affine> local(count), all(count * pr->world);
        for (size_t i = 0; i < count; ++i)
            PK_TRY(zk_msm_g1(ctx, (const Fr*)ptrs[i] + lo, bases + lo, len, &local[i]));
        if (pr->gather(pr->gather_user, local.data(), count * sizeof(G1Affine), all.data())) return ctx->fail(ZK_ERR_INVALID_ARG, "sharded session: all-gather callback failed");
        std::vector<G1Affine> parts(pr->world);
        for (size_t i = 0; i < count; ++i) {
            for (uint32_t q_ = 0; q_ < pr->world; ++q_) parts[q_] = all[(size_t)q_ * count + i];
            if (zk_g1_sum_host(parts.data(), pr->world, &out[i])) return ctx->fail(ZK_ERR_INVALID_ARG, "sharded session: partial sums could not be added");
        }
        return ZK_OK;
    }
    const size_t per = (count + pr->world - 1) / pr->world;
    std::vector<const void*> mine;
    for (size_t i = pr->rank; i < count; i += pr->world) mine.push_back(ptrs[i]);
    std::vector<G1Affine> local(per), all(per * pr->world);
    memset((void*)local.data(), 0, sizeof(G1Affine) * per);
    PK_TRY(commit_batch_staged(ctx, srs, basis, mine.data(), mine.size(), n, local.data(), nullptr, nullptr, hint.data()));
    if (per && pr->gather(pr->gather_user, local.data(), per * sizeof(G1Affine), all.data())) return ctx->fail(ZK_ERR_INVALID_ARG, "sharded session: all-gather callback failed");
    for (size_t i = 0; i < count; ++i) out[i] = all[(i % pr->world) * per + i / pr->world];
    return ZK_OK;
}

int upload(zk_ctx* ctx, DevBuf* b, const void* h, size_t bytes) {
    if (!b->alloc(bytes)) return ctx->fail(ZK_ERR_OOM, "prover: alloc of %zu bytes failed", bytes);
    return zk_h2d(ctx, b->p, h, bytes);
}

}  // namespace

extern "C" {

void zk_pk_destroy(zk_ctx* ctx, zk_pk* pk) {
    if (ctx) (void)zk_ctx_sync(ctx);
    if (ctx && pk) ctx->coset_cache_bytes -= std::min(ctx->coset_cache_bytes, pk->part_cache_bytes);     // its slots go back to the shared budget
    delete pk;
}

// The constraint-system part of a key blob (everything before the column data) into `pk`: shape, phases, query lists,
// permutation columns, constants, gate and lookup programs -- with every count checked against what the blob can hold and the
// declared degree against what the programs need.  No device, no SRS (zk_pk_create goes on from here; the host-only hooks stop here).
static int parse_cs(Reader& r, zk_pk* pk, size_t blob_len, bool with_columns, std::string* err) {
    char msg[256];
    auto fail = [&](int code, const char* fmt, auto... args) { snprintf(msg, sizeof msg, fmt, args...); *err = msg; return code; };
    if (r.u32() != 0x4B505A4Bu) return fail(ZK_ERR_INVALID_ARG, "pk blob: bad magic");
    const uint32_t version = r.u32();
    if (version != 3u) return fail(ZK_ERR_INVALID_ARG, "pk blob: unsupported version %u (this library reads version 3)", version);
    pk->k = r.u32(); pk->bf = r.u32(); pk->d = r.u32(); pk->F = r.u32(); pk->A = r.u32(); pk->I = r.u32(); pk->P = r.u32(); pk->L = r.u32();
    const uint32_t ngates = r.u32(), nconsts = r.u32();
    // halo2: cs.degree() >= 3 (the permutation argument); the extended domain has at most 2^28 rows
    if (!r.ok || pk->k < 2 || pk->k > 27 || pk->d < 3 || pk->d > 17) return fail(ZK_ERR_INVALID_ARG, "pk blob: bad header (k=%u, degree=%u)", pk->k, pk->d);
    const size_t n = (size_t)1 << pk->k;
    if (pk->bf + 2 >= n) return fail(ZK_ERR_INVALID_ARG, "pk blob: too many blinding rows");
    // every count of the header is checked against what the blob can hold before anything is sized by it
    if ((size_t)pk->A * 4 > r.left || (size_t)pk->P * 8 > r.left || (size_t)nconsts * 32 > r.left || (size_t)ngates * 4 > r.left || (size_t)pk->L * 8 > r.left ||
        (with_columns && ((size_t)pk->F + pk->P) > r.left / (n * 32)) || pk->A > 0xFFFFFFu || pk->F > 0xFFFFFFu || pk->I > 0xFFFFFFu)
        return fail(ZK_ERR_INVALID_ARG, "pk blob: header counts exceed the blob (%zu bytes)", blob_len);
    pk->u = (uint32_t)n - pk->bf - 1;
    pk->chunk = pk->d - 2;
    pk->C = pk->P ? (pk->P + pk->chunk - 1) / pk->chunk : 0;
    pk->ext_k = pk->k;
    while (((size_t)1 << pk->ext_k) < n * (pk->d - 1)) ++pk->ext_k;
    if (pk->ext_k > 28) return fail(ZK_ERR_INVALID_ARG, "pk blob: extended domain 2^%u exceeds the two-adicity of Fr", pk->ext_k);
    pk->adv_phase.assign(pk->A, 0);
    {   // phases: [num_challenges][A x advice phase][num_challenges x challenge phase]
        const uint32_t nch = r.count(4);
        if (nch > 4096) return fail(ZK_ERR_INVALID_ARG, "pk blob: too many challenges");
        for (uint32_t i = 0; i < pk->A && r.ok; ++i) { pk->adv_phase[i] = r.u32(); if (pk->adv_phase[i] + 1 > pk->num_phases) pk->num_phases = pk->adv_phase[i] + 1; }
        for (uint32_t i = 0; i < nch && r.ok; ++i) { pk->chal_phase.push_back(r.u32()); if (pk->chal_phase[i] + 1 > pk->num_phases) pk->num_phases = pk->chal_phase[i] + 1; }
        if (!r.ok || pk->num_phases > 16) return fail(ZK_ERR_INVALID_ARG, "pk blob: bad phase table");
    }
    // cs.advice_queries / fixed_queries / instance_queries: (column, rotation) in registration order
    r.queries(&pk->adv_q, CT_ADVICE, pk->A);
    r.queries(&pk->fix_q, CT_FIXED, pk->F);
    r.queries(&pk->inst_q, CT_INSTANCE, pk->I);
    if (!r.ok) return fail(ZK_ERR_INVALID_ARG, "pk blob: bad query lists");
    for (uint32_t i = 0; i < pk->P && r.ok; ++i) { uint32_t t = r.u32(), x = r.u32(); pk->perm_cols.push_back({t, x}); }
    for (uint32_t i = 0; i < nconsts && r.ok; ++i) { const uint8_t* b = r.bytes(32); F4 v; if (b) memcpy(v.l, b, 32); pk->consts.push_back(v); }
    for (uint32_t i = 0; i < ngates && r.ok; ++i) pk->gates.push_back(r.prog());
    for (uint32_t i = 0; i < pk->L && r.ok; ++i) {
        zk_pk::Lookup lk;
        const uint32_t m = r.u32(), ninputs = r.count(4);
        if (!r.ok || m == 0 || ninputs == 0 || (size_t)m * 4 > r.left || (size_t)m * ninputs > r.left / 4) { r.ok = false; break; }
        for (uint32_t j = 0; j < m && r.ok; ++j) lk.tables.push_back(r.prog());
        for (uint32_t a = 0; a < ninputs && r.ok; ++a) {
            lk.inputs.emplace_back();
            for (uint32_t j = 0; j < m && r.ok; ++j) lk.inputs.back().push_back(r.prog());
        }
        pk->lookups.push_back(std::move(lk));
    }
    if (!r.ok) return fail(ZK_ERR_INVALID_ARG, "pk blob: truncated or malformed constraint system");
    // The degree the blob declares must cover what its own programs need (halo2 ConstraintSystem::degree:
    // permutation argument 3, mv_lookup::Argument::required_degree, every gate polynomial): with a
    // smaller one the quotient would not fit its d - 1 pieces and the proof would be rejected.
    {
        uint32_t need = 3;
        std::vector<int> tmp_deg;
        for (const Prog& g : pk->gates) {
            const int dg = program_degree(g, &tmp_deg);
            if (dg < 0) return fail(ZK_ERR_INVALID_ARG, "pk blob: malformed gate program");
            need = std::max(need, (uint32_t)dg);
        }
        for (const auto& lk : pk->lookups) {
            std::vector<int> none;
            int table_degree = 0, inputs_degree = 0;
            for (const Prog& g : lk.tables) { const int dg = program_degree(g, &none); if (dg < 0) return fail(ZK_ERR_INVALID_ARG, "pk blob: malformed lookup table program"); table_degree = std::max(table_degree, dg); }
            for (const auto& in : lk.inputs) {
                int one = 0;
                for (const Prog& g : in) { const int dg = program_degree(g, &none); if (dg < 0) return fail(ZK_ERR_INVALID_ARG, "pk blob: malformed lookup input program"); one = std::max(one, dg); }
                inputs_degree += one;
            }
            need = std::max(need, std::max((uint32_t)(3 + lk.inputs.size()), (uint32_t)(table_degree + inputs_degree + 2)));
        }
        if (pk->d < need) return fail(ZK_ERR_INVALID_ARG, "pk blob: declared degree %u is below the %u its gates and lookup arguments require", pk->d, need);
    }
    for (const auto& pc : pk->perm_cols) {          // halo2: enable_equality queries the column at Rotation::cur()
        if (pc.first > CT_INSTANCE) return fail(ZK_ERR_INVALID_ARG, "pk blob: bad permutation column type %u", pc.first);
        if (pc.first == CT_INSTANCE) continue;
        const std::vector<Query>& qs = pc.first == CT_ADVICE ? pk->adv_q : pk->fix_q;
        bool seen = false;
        for (const Query& q : qs) if (q.idx == pc.second && q.rot == 0) { seen = true; break; }
        if (!seen) return fail(ZK_ERR_INVALID_ARG, "pk blob: permutation column (%u, %u) is not queried at rotation 0", pc.first, pc.second);
    }
    return ZK_OK;
}

// keygen_pk: parse the blob, make the key material device resident, commit fixed / sigma columns.
int zk_pk_create(zk_ctx* ctx, const zk_srs* srs, const void* h_blob, size_t blob_len, zk_pk** out) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, srs && h_blob && out, "null pointer");
    Reader r{(const uint8_t*)h_blob, blob_len};
    std::unique_ptr<zk_pk> pk(new zk_pk());
    pk->srs = srs;
    {
        std::string err;
        const int rc = parse_cs(r, pk.get(), blob_len, true, &err);
        if (rc) return ctx->fail(rc, "%s", err.c_str());
    }
    // commit_lagrange needs the Lagrange basis of exactly this domain (halo2: ParamsKZG::downsize)
    if (pk->k != srs->k) return ctx->fail(ZK_ERR_INVALID_ARG, "SRS is for k=%u but the circuit has k=%u: downsize the SRS first", srs->k, pk->k);
    if (!srs->g_lagrange) return ctx->fail(ZK_ERR_INVALID_ARG, "SRS has no Lagrange basis");
    const size_t n = (size_t)1 << pk->k;
    const size_t cs_len = blob_len - r.left;        // the constraint-system part: everything before the column data
    // fixed + sigma columns
    pk->fixed_lag.resize(pk->F); pk->fixed_coeff.resize(pk->F);
    pk->sigma_lag.resize(pk->P); pk->sigma_coeff.resize(pk->P);
    pk->fixed_com.resize(pk->F); pk->sigma_com.resize(pk->P);
    {   // keygen's commit_lagrange over every fixed and sigma column as ONE pipelined batch: column i + 1
        // crosses PCIe on the copy stream while the MSM of column i runs; the coefficient forms follow
        std::vector<const void*> h_cols;
        std::vector<void*> d_cols;
        for (uint32_t i = 0; i < pk->F + pk->P; ++i) {
            const uint8_t* b = r.bytes(n * 32);
            if (!b) return ctx->fail(ZK_ERR_INVALID_ARG, "pk blob: truncated %s column", i < pk->F ? "fixed" : "sigma");
            DevBuf& lag = i < pk->F ? pk->fixed_lag[i] : pk->sigma_lag[i - pk->F];
            if (!lag.alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc of %zu bytes failed", n * 32);
            h_cols.push_back(b);
            d_cols.push_back(lag.p);
        }
        std::vector<G1Affine> coms(h_cols.size());
        PK_TRY(zk_commit_batch_h2d(ctx, srs, 1, h_cols.data(), d_cols.data(), h_cols.size(), n, coms.data()));
        for (uint32_t i = 0; i < pk->F; ++i) { pk->fixed_com[i] = coms[i]; PK_TRY(to_coeff(ctx, pk.get(), pk->fixed_lag[i], &pk->fixed_coeff[i])); }
        for (uint32_t i = 0; i < pk->P; ++i) { pk->sigma_com[i] = coms[pk->F + i]; PK_TRY(to_coeff(ctx, pk.get(), pk->sigma_lag[i], &pk->sigma_coeff[i])); }
    }
    // l0, l_last, l_active (built on the device: a delta at row 0, a delta at row u, ones below u) and the omega^i column
    {
        const Fr w = fr_root_of_unity(pk->k), one = Fr::one();
        if (!pk->l0_lag.alloc(n * 32) || !pk->llast_lag.alloc(n * 32) || !pk->lactive_lag.alloc(n * 32) || !pk->omega_lag.alloc(n * 32))
            return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
        ZK_HIP(ctx, hipMemsetAsync(pk->l0_lag.p, 0, n * 32, ctx->stream));
        ZK_HIP(ctx, hipMemsetAsync(pk->llast_lag.p, 0, n * 32, ctx->stream));
        ZK_HIP(ctx, hipMemsetAsync(pk->lactive_lag.p, 0, n * 32, ctx->stream));
        PK_TRY(zk_h2d(ctx, pk->l0_lag.p, &one, 32));
        PK_TRY(zk_h2d(ctx, (char*)pk->llast_lag.p + (size_t)pk->u * 32, &one, 32));
        PK_TRY(zk_fr_powers(ctx, &one, &one, pk->lactive_lag.p, pk->u));          // 1 * 1^i: ones on the usable rows
        PK_TRY(to_coeff(ctx, pk.get(), pk->l0_lag, &pk->l0_coeff));
        PK_TRY(to_coeff(ctx, pk.get(), pk->llast_lag, &pk->llast_coeff));
        PK_TRY(to_coeff(ctx, pk.get(), pk->lactive_lag, &pk->lactive_coeff));
        PK_TRY(zk_fr_powers(ctx, &w, &one, pk->omega_lag.p, n));
        PK_TRY(zk_ctx_sync(ctx));
    }
    // Default vk.transcript_repr: Blake2b-512 ("Halo2-Verify-Key") over the whole constraint-system
    // part of the blob and the compressed fixed / sigma commitments.  halo2's own value hashes the Debug
    // string of the pinned verifying key, which only the Rust side can produce: the shim installs it
    // with zk_pk_set_transcript_repr, and then proofs are made for exactly the reference's verifier.
    {
        host::Blake2b hsh;
        hsh.init("Halo2-Verify-Key");
        hsh.update(h_blob, cs_len);
        for (const auto& c : pk->fixed_com) { uint8_t b[32]; host::g1_compress(c, b); hsh.update(b, 32); }
        for (const auto& c : pk->sigma_com) { uint8_t b[32]; host::g1_compress(c, b); hsh.update(b, 32); }
        uint8_t dg[64];
        hsh.finalize(dg);
        pk->vk_repr = host::fr_from_uniform(dg);
    }
    PK_TRY(zk_ctx_sync(ctx));
    *out = pk.release();
    return ZK_OK;
}

// halo2 absorbs `vk.transcript_repr()` first [REF zkevm-circuits/src/super_circuit/test.rs:70-85 pins its
// value for the SuperCircuit]; the caller that holds the real VerifyingKey passes that scalar here.
int zk_pk_set_transcript_repr(zk_ctx* ctx, zk_pk* pk, const void* h_repr) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pk && h_repr, "null pointer");
    memcpy(pk->vk_repr.l, h_repr, 32);
    return ZK_OK;
}

// Shape of a key, for callers that size buffers from it: out[0..15] = k, degree, extended k, F, A, I,
// P (permutation columns), C (permutation chunks), L (lookup arguments), phases, challenges,
// blinding factors, advice queries, fixed queries, commitments in a proof, evaluations in a proof.
int zk_pk_shape(zk_ctx* ctx, const zk_pk* pk, uint32_t* out16) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pk && out16, "null pointer");
    const uint32_t evals = (uint32_t)pk->adv_q.size() + (uint32_t)pk->fix_q.size() + 1 + pk->P + (pk->C ? 3 * pk->C - 1 : 0) + 3 * pk->L;
    const uint32_t v[16] = {pk->k, pk->d, pk->ext_k, pk->F, pk->A, pk->I, pk->P, pk->C, pk->L, pk->num_phases, (uint32_t)pk->chal_phase.size(), pk->bf,
                            (uint32_t)pk->adv_q.size(), (uint32_t)pk->fix_q.size(), pk->A + 2 * pk->L + pk->C + 1 + (pk->d - 1), evals};
    memcpy(out16, v, sizeof v);
    return ZK_OK;
}

// vk side of the key: commitments (F fixed then P sigma, 64-byte affine each) and vk_repr
int zk_pk_vk(zk_ctx* ctx, const zk_pk* pk, void* h_commitments, void* h_vk_repr) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pk, "null pointer");
    if (h_commitments) {
        G1Affine* o = (G1Affine*)h_commitments;
        for (uint32_t i = 0; i < pk->F; ++i) o[i] = pk->fixed_com[i];
        for (uint32_t i = 0; i < pk->P; ++i) o[pk->F + i] = pk->sigma_com[i];
    }
    if (h_vk_repr) memcpy(h_vk_repr, &pk->vk_repr, 32);
    return ZK_OK;
}

// ---- proving session: begin -> one call per advice phase -> finish ---------------------------------
// halo2's create_proof synthesises the circuit once per phase and squeezes that phase's challenges
// after committing its advice columns (the SuperCircuit has three phases
// [REF zkevm-circuits/src/util.rs:120-133]); witness synthesis stays on the host, so the phases
// are separate calls and each returns the challenges the next synthesis pass needs.
void zk_proof_abort(zk_ctx* ctx, zk_proof* pr) {
    if (ctx) (void)zk_ctx_sync(ctx);
    delete pr;
}

int zk_proof_set_multiopen(zk_ctx* ctx, zk_proof* pr, int kind) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pr && (kind == ZK_MULTIOPEN_GWC || kind == ZK_MULTIOPEN_SHPLONK), "unknown multi-open scheme");
    pr->multiopen = kind;
    return ZK_OK;
}

int zk_proof_set_vanishing_random(zk_ctx* ctx, zk_proof* pr, int kind) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pr && (kind == ZK_VANISHING_UNIFORM || kind == ZK_VANISHING_ONE), "unknown kind of vanishing-argument polynomial");
    pr->vanishing_random = kind;
    return ZK_OK;
}

int zk_proof_set_transcript(zk_ctx* ctx, zk_proof* pr, const zk_transcript_vtable* vt, void* user) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pr && vt && vt->common_point && vt->common_scalar && vt->write_point && vt->write_scalar && vt->squeeze_challenge, "incomplete transcript vtable");
    ZK_REQUIRE(ctx, pr->phase == 0 && !pr->tr.vt, "the transcript must be set once, before the first advice phase");
    pr->ext_vt = *vt;
    pr->tr.vt = &pr->ext_vt;
    pr->tr.user = user;
    for (const F4& s_ : pr->absorbed) pr->tr.common_scalar(s_);
    pr->absorbed.clear();
    pr->absorbed.shrink_to_fit();
    if (pr->tr.err) return ctx->fail(ZK_ERR_INVALID_ARG, "external transcript callback failed with status %d", pr->tr.err);
    return ZK_OK;
}

// Built-in transcripts: Blake2b (default), Poseidon (gen_snark_shplonk) or Keccak / EVM
// (gen_evm_proof_shplonk).  Same rule as zk_proof_set_transcript: right after zk_proof_begin; what
// begin absorbed is replayed into the chosen transcript.
int zk_proof_set_transcript_kind(zk_ctx* ctx, zk_proof* pr, int kind) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pr && (kind == ZK_TRANSCRIPT_BLAKE2B || kind == ZK_TRANSCRIPT_POSEIDON || kind == ZK_TRANSCRIPT_EVM), "unknown transcript kind");
    ZK_REQUIRE(ctx, pr->phase == 0 && !pr->tr.vt, "the transcript must be chosen once, before the first advice phase");
    pr->tr.reset(kind);
    for (const F4& s_ : pr->absorbed) pr->tr.common_scalar(s_);
    return ZK_OK;
}

int zk_proof_set_sharding(zk_ctx* ctx, zk_proof* pr, uint32_t rank, uint32_t world, zk_allgather_fn gather, void* user) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pr && world >= 1 && rank < world && (world == 1 || gather), "need rank < world and an all-gather callback");
    ZK_REQUIRE(ctx, pr->phase == 0, "sharding must be set before the first advice phase");
    pr->rank = rank; pr->world = world; pr->gather = gather; pr->gather_user = user;
    return ZK_OK;
}

// Sharded session over the context's own RCCL communicator (zk_comm_init): rank and world come from it,
// commitments are all-gathered through a device staging buffer, advice columns and finished quotient
// cosets device to device over xGMI -- no callback, no host-language collective needed.
static int comm_gather_host_thunk(void* user, const void* send, size_t bytes, void* recv) { return comm_allgather_host((zk_ctx*)user, send, bytes, recv); }
static int comm_gather_dev_thunk(void* user, const void* d_send, size_t bytes, void* d_recv) {
    zk_ctx* c = (zk_ctx*)user;
    const int rc = comm_allgather_dev(c, d_send, bytes, d_recv);
    return rc ? rc : (hipStreamSynchronize(c->stream) == hipSuccess ? 0 : ZK_ERR_HIP);     // the callback contract: complete on return
}
int zk_proof_set_sharding_comm(zk_ctx* ctx, zk_proof* pr) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pr, "null pointer");
    ZK_REQUIRE(ctx, comm_ready(ctx), "no communicator on this context: call zk_comm_init first");
    ZK_REQUIRE(ctx, pr->phase == 0, "sharding must be set before the first advice phase");
    pr->rank = ctx->comm_rank; pr->world = ctx->comm_world;
    pr->gather = comm_gather_host_thunk; pr->gather_user = ctx;
    pr->gather_dev = comm_gather_dev_thunk; pr->gather_dev_user = ctx;
    pr->use_comm = true;
    return ZK_OK;
}

int zk_proof_set_device_gather(zk_ctx* ctx, zk_proof* pr, zk_allgather_fn gather_dev, void* user) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pr && gather_dev, "null pointer");
    ZK_REQUIRE(ctx, pr->world > 1 && pr->gather, "set the sharding (zk_proof_set_sharding) first");
    ZK_REQUIRE(ctx, pr->phase == 0, "must be set before the first advice phase");
    pr->gather_dev = gather_dev;
    pr->gather_dev_user = user;
    return ZK_OK;
}

// create_proof's instance handling: every PROVIDED value is absorbed (KZG: instances are not
// committed), a column longer than the usable rows is Error::InstanceTooLarge, and the column is
// zero-padded to n.  h_len == NULL is the full-column form: the usable rows of an n-row image.
static int proof_begin(zk_ctx* ctx, const zk_pk* pk, const void* const* h_instance, const uint32_t* h_len, const uint8_t* seed16, zk_proof** out) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    PoolScope pool_scope(ctx);
    ZK_REQUIRE(ctx, pk && seed16 && out && (h_instance || !pk->I), "null pointer");
    const size_t n = (size_t)1 << pk->k;
    if (h_len)
        for (uint32_t i = 0; i < pk->I; ++i)
            if (h_len[i] > pk->u) return ctx->fail(ZK_ERR_INVALID_ARG, "instance column %u has %u values, the circuit has %zu usable rows (InstanceTooLarge)", i, h_len[i], pk->u);
    std::unique_ptr<zk_proof> pr(new zk_proof(pk, seed16));
    pr->tr.common_scalar(pk->vk_repr);
    pr->absorbed.push_back(pk->vk_repr);
    for (uint32_t i = 0; i < pk->I; ++i) {
        const F4* v = (const F4*)h_instance[i];
        const size_t len = h_len ? h_len[i] : pk->u;
        ZK_REQUIRE(ctx, v || !len, "null instance column");
        for (size_t row = 0; row < len; ++row) { pr->tr.common_scalar(v[row]); pr->absorbed.push_back(v[row]); }
        if (h_len) {
            if (!pr->inst_lag[i].alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc of %zu bytes failed", n * 32);
            ZK_HIP(ctx, hipMemsetAsync(pr->inst_lag[i].p, 0, n * 32, ctx->stream));
            if (len) PK_TRY(zk_h2d(ctx, pr->inst_lag[i].p, v, len * 32));
        } else {
            PK_TRY(upload(ctx, &pr->inst_lag[i], h_instance[i], n * 32));
        }
        PK_TRY(to_coeff(ctx, pk, pr->inst_lag[i], &pr->inst_coeff[i]));
    }
    *out = pr.release();
    return ZK_OK;
}
int zk_proof_begin(zk_ctx* ctx, const zk_pk* pk, const void* const* h_instance, const uint8_t* seed16, zk_proof** out) {
    return proof_begin(ctx, pk, h_instance, nullptr, seed16, out);
}
int zk_proof_begin_instances(zk_ctx* ctx, const zk_pk* pk, const void* const* h_instance, const uint32_t* h_instance_len, const uint8_t* seed16, zk_proof** out) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pk && (h_instance_len || !pk->I), "null pointer");
    static const uint32_t none = 0;
    return proof_begin(ctx, pk, h_instance, h_instance_len ? h_instance_len : &none, seed16, out);
}

// Commits the advice columns of the current phase (h_cols[j] is advice column col_index[j]; exactly
// the columns of this phase, each n x 32 B Lagrange values; rows >= n - bf are replaced by blinding
// values) and squeezes the challenges that become available after it into h_challenges (Fr each,
// in challenge-index order).  When h_challenges is given, *num_challenges holds its capacity (in
// challenges) on entry; on return it holds how many the phase produced.
static int plan_advice_cosets(zk_ctx* ctx, zk_proof* pr);
static int advice_lagrange_readers(zk_ctx* ctx, const zk_pk* pk, std::vector<uint8_t>* need);
static int advice_phase_impl(zk_ctx* ctx, zk_proof* pr, const uint32_t* col_index, const void* const* h_cols, uint32_t ncols, void* h_challenges, uint32_t* num_challenges, bool dev_src, bool in_place);
int zk_proof_advice_phase(zk_ctx* ctx, zk_proof* pr, const uint32_t* col_index, const void* const* h_cols, uint32_t ncols, void* h_challenges, uint32_t* num_challenges) {
    return advice_phase_impl(ctx, pr, col_index, h_cols, ncols, h_challenges, num_challenges, false, false);
}
// The same phase for witness columns that are RESIDENT ON THE DEVICE (n x 32 B each, Montgomery, device pointers): a witness
// generated on the GPU, or one uploaded ahead of the proof.  Judged small / dense on the device.  Same transcript, same bytes as the
// host-column call.  flags:
//   0                        the columns are copied (device to device, on the copy stream) into the session's own buffers; the caller's
//                            stay untouched
//   ZK_ADVICE_DEV_IN_PLACE   the session works IN the caller's buffers: it overwrites their last blinding_factors + 1 rows (the rows
//                            halo2 fills with blinding values; a witness has nothing there) and reads them until zk_proof_finish /
//                            zk_proof_abort returns.  No copy, and n x 32 B per column less device memory -- which is what lets a
//                            1000-column session keep two cosets of every column computed ahead.
int zk_proof_advice_phase_dev(zk_ctx* ctx, zk_proof* pr, const uint32_t* col_index, const void* const* d_cols, uint32_t ncols, uint32_t flags, void* h_challenges, uint32_t* num_challenges) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, (flags & ~(uint32_t)ZK_ADVICE_DEV_IN_PLACE) == 0, "unknown flag");
    return advice_phase_impl(ctx, pr, col_index, d_cols, ncols, h_challenges, num_challenges, true, (flags & ZK_ADVICE_DEV_IN_PLACE) != 0);
}
static int advice_phase_impl(zk_ctx* ctx, zk_proof* pr, const uint32_t* col_index, const void* const* h_cols, uint32_t ncols, void* h_challenges, uint32_t* num_challenges, bool dev_src, bool in_place) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    PoolScope pool_scope(ctx);
    ZK_REQUIRE(ctx, pr && (ncols == 0 || (col_index && h_cols)), "null pointer");
    const zk_pk* pk = pr->pk;
    if (pr->phase >= pk->num_phases) return ctx->fail(ZK_ERR_INVALID_ARG, "all %u advice phases are already committed", pk->num_phases);
    const size_t n = (size_t)1 << pk->k;
    uint32_t expected = 0, phase_challenges = 0;
    for (uint32_t i = 0; i < pk->A; ++i) expected += pk->adv_phase[i] == pr->phase;
    for (uint32_t cp : pk->chal_phase) phase_challenges += cp == pr->phase;
    if (h_challenges && (!num_challenges || *num_challenges < phase_challenges))
        return ctx->fail(ZK_ERR_INVALID_ARG, "phase %u yields %u challenges: pass a buffer for at least that many and its capacity in *num_challenges", pr->phase, phase_challenges);
    if (ncols != expected) return ctx->fail(ZK_ERR_INVALID_ARG, "phase %u has %u advice columns, %u were passed", pr->phase, expected, ncols);
    // Sharded session with a device all-gather and a device-resident witness: a rank holds only the columns it OWNS (position j of
    // the phase's columns in ascending column order, j % world == rank); the others arrive over the fabric and may be passed as NULL.
    const bool owner_only = dev_src && pr->world > 1 && pr->gather && pr->gather_dev;
    std::vector<const void*> by_col(pk->A, nullptr);
    std::vector<uint8_t> seen(pk->A, 0);
    for (uint32_t j = 0; j < ncols; ++j) {
        const uint32_t c = col_index[j];
        if (c >= pk->A || pk->adv_phase[c] != pr->phase || seen[c] || (!h_cols[j] && !owner_only)) return ctx->fail(ZK_ERR_INVALID_ARG, "advice column %u does not belong to phase %u (or is repeated)", c, pr->phase);
        by_col[c] = h_cols[j];
        seen[c] = 1;
    }
    if (owner_only) {
        uint32_t pos = 0;
        for (uint32_t c = 0; c < pk->A; ++c) {
            if (!seen[c]) continue;
            if (pos % pr->world == pr->rank && !by_col[c]) return ctx->fail(ZK_ERR_INVALID_ARG, "advice column %u is owned by this rank (position %u of the phase, rank %u of %u) and was passed as NULL", c, pos, pr->rank, pr->world);
            ++pos;
        }
    }
    if (dev_src) pr->advice_on_device = true;
    if (in_place) {
        std::vector<const void*> sorted_ptrs;
        for (uint32_t j = 0; j < ncols; ++j) if (h_cols[j]) sorted_ptrs.push_back(h_cols[j]);
        std::sort(sorted_ptrs.begin(), sorted_ptrs.end());
        if (std::adjacent_find(sorted_ptrs.begin(), sorted_ptrs.end()) != sorted_ptrs.end()) return ctx->fail(ZK_ERR_INVALID_ARG, "in-place witness columns must be distinct buffers (each gets its own blinding rows)");
        pr->advice_in_place = true;
    }
    StageTrace trace(ctx);
    // Column c+1 is uploaded on the copy stream while the MSM of column c runs; the last bf + 1 rows
    // (u .. n - 1) of every column are blinding values (drawn up front, in column-index = transcript order).
    struct Stage {
        zk_ctx* ctx; size_t body, tail;
        std::vector<const void*> src; std::vector<void*> dst; std::vector<F4> blind_v; const F4* blind = nullptr;
        uint32_t world = 1;
        hipMemcpyKind kind = hipMemcpyHostToDevice;                       // device-resident witness: device to device
        std::vector<size_t> own;                                         // device-gather mode: only these columns are uploaded by this rank
        const zk_pk* pk = nullptr; std::vector<DevBuf*> lag, coeff;     // coefficient forms are produced as the columns arrive
        // plain (unsharded) sessions: coefficient forms AND the planned cosets of the columns, several columns per launch, on the
        // auxiliary stream behind their uploads
        zk_proof* pr = nullptr; std::vector<uint32_t> col; std::vector<size_t> pending; std::vector<Fr> g_of_r;
        int flush() {
            if (pending.empty()) return ZK_OK;
            hipStream_t main_stream = ctx->stream;
            ctx->stream = ctx->stream_aux;
            struct Back { zk_ctx* c; hipStream_t s; ~Back() { c->stream = s; } } back{ctx, main_stream};
            const size_t n_ = (size_t)1 << pk->k;
            std::vector<Fr*> dsts;
            std::vector<const Fr*> srcs;
            for (size_t c_ : pending) {
                if (!coeff[c_]->alloc(n_ * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
                dsts.push_back(coeff[c_]->fr());
                srcs.push_back(lag[c_]->fr());
            }
            const Fr omega_inv = fr_inv_host(fr_root_of_unity(pk->k)), ninv = fr_inv_host(fr_from_u64(1ull << pk->k));
            PK_TRY(ntt_run_many(ctx, dsts.data(), srcs.data(), dsts.size(), pk->k, omega_inv, &ninv, nullptr, nullptr, false));
            for (size_t r = 0; r < g_of_r.size(); ++r) {
                std::vector<const void*> csrc;
                std::vector<void*> cdst;
                for (size_t c_ : pending) {
                    const uint32_t gc = col[c_];
                    if (!(pr->pre_mask[gc] >> r & 1u)) continue;
                    DevBuf& slot = pr->adv_coset[r][gc];
                    if (!slot.alloc(n_ * 32)) {          // the estimate was too generous: stop computing ahead, the quotient transforms the rest itself
                        for (uint32_t& m_ : pr->pre_mask) m_ = 0;
                        break;
                    }
                    csrc.push_back(coeff[c_]->p);
                    cdst.push_back(slot.p);
                }
                if (!csrc.empty()) PK_TRY(zk_coeff_to_coset_batch(ctx, csrc.data(), pk->k, &g_of_r[r], cdst.data(), csrc.size()));
            }
            pending.clear();
            return ZK_OK;
        }
    } sg{ctx, (size_t)pk->u * 32, (size_t)(pk->bf + 1) * 32, {}, {}, {}};   // halo2: advice_values[n - (blinding_factors + 1)..] are random, row u included
    sg.pk = pk;
    sg.pr = pr;
    sg.kind = dev_src ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    for (uint32_t c = 0; c < pk->A; ++c) {        // column-index order = transcript order
        if (!seen[c]) continue;
        if (in_place && by_col[c] && (!owner_only || sg.dst.size() % pr->world == pr->rank)) pr->adv_lag[c].borrow(const_cast<void*>(by_col[c]));
        else if (!pr->adv_lag[c].alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc of %zu bytes failed", n * 32);
        sg.src.push_back(by_col[c]);
        sg.dst.push_back(pr->adv_lag[c].p);
        sg.lag.push_back(&pr->adv_lag[c]);
        sg.coeff.push_back(&pr->adv_coeff[c]);
        sg.col.push_back(c);
        for (uint32_t r = 0; r <= pk->bf; ++r) sg.blind_v.push_back(pr->rng.next_fr());
    }
    PinnedBuf blind_pinned;
    if (!blind_pinned.alloc(sg.blind_v.size() * sizeof(F4))) return ctx->fail(ZK_ERR_OOM, "prover: pinned staging allocation failed");
    memcpy(blind_pinned.p, sg.blind_v.data(), sg.blind_v.size() * sizeof(F4));
    sg.blind = (const F4*)blind_pinned.p;
    trace.mark("  advice: blinding rows drawn and staged");
    PK_TRY(copy_stream_open(ctx));
    if (!ctx->ensure_aux()) return ctx->fail(ZK_ERR_HIP, "could not create the auxiliary stream");
    ZK_HIP(ctx, hipEventRecord(ctx->ev_aux, ctx->stream));                 // pooled blocks: everything enqueued so far comes first
    ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream_aux, ctx->ev_aux, 0));
    // Sharded session: every rank uploads every column (each GPU has its own PCIe link; all of them
    // are needed for the quotient) but commits only columns rank, rank + world, ...: the upload of
    // `world` columns hides one MSM.
    sg.world = pr->world > 1 && pr->gather ? pr->world : 1;
    if (!pr->pre_planned) PK_TRY(plan_advice_cosets(ctx, pr));
    trace.mark("  advice: coset plan");
    if (sg.world == 1 && !pr->adv_coset.empty()) {
        const Fr w_ext = fr_root_of_unity(pk->ext_k);
        Fr g = fr_zeta();
        for (size_t r = 0; r < pr->adv_coset.size(); ++r) { sg.g_of_r.push_back(g); g = g * w_ext; }
    }
    auto stage = [](void* user, size_t it) -> int {
        Stage* s_ = (Stage*)user;
        if (!s_->own.empty()) {      // device-gather mode: one own column per call
            if (it > 0) PK_TRY(to_coeff_aux(s_->ctx, s_->pk, *s_->lag[s_->own[it - 1]], s_->coeff[s_->own[it - 1]]));
            const size_t c_ = s_->own[it];
            if (s_->dst[c_] != s_->src[c_]) ZK_HIP(s_->ctx, hipMemcpyAsync(s_->dst[c_], s_->src[c_], s_->body, s_->kind, s_->ctx->stream_copy));
            ZK_HIP(s_->ctx, hipMemcpyAsync((char*)s_->dst[c_] + s_->body, s_->blind + c_ * (s_->tail / 32), s_->tail, hipMemcpyHostToDevice, s_->ctx->stream_copy));
            PK_TRY(copy_stream_fence(s_->ctx));
            ZK_HIP(s_->ctx, hipStreamWaitEvent(s_->ctx->stream_aux, s_->ctx->ev_copy, 0));
            return ZK_OK;
        }
        // the group uploaded by the previous call is on the device (the main stream has waited for it):
        // its lagrange_to_coeff runs now, on a main stream that is otherwise waiting for PCIe
        if (it > 0 && s_->world == 1) {
            s_->pending.push_back(it - 1);
            static const size_t flush_at = [] { const char* e = getenv("ZK_ADVICE_NTT_GROUP"); const int v = e ? atoi(e) : 32; return (size_t)(v < 1 ? 1 : v > 64 ? 64 : v); }();      // measurement knob
            if (s_->pending.size() >= flush_at) PK_TRY(s_->flush());        // two launch pairs of sixteen columns (ntt_run_many's granularity at k = 20) per hand-over to the auxiliary stream: headline 0.9785 / 0.9882 s with 16, 0.9738 / 0.9851 with 32, 0.9812 with 64, 0.9788 with 8 (two alternating A/B runs)
        } else if (it > 0)
            for (size_t c_ = (it - 1) * s_->world; c_ < std::min(it * (size_t)s_->world, s_->dst.size()); ++c_)
                PK_TRY(to_coeff_aux(s_->ctx, s_->pk, *s_->lag[c_], s_->coeff[c_]));
        static const bool skip_upload = getenv("ZK_DEBUG_SKIP_UPLOAD") != nullptr;       // measurement only (the proof is garbage): is the phase bound by PCIe or by the device?
        for (size_t c_ = it * s_->world; c_ < std::min((it + 1) * (size_t)s_->world, s_->dst.size()); ++c_) {
            if (skip_upload) continue;
            if (s_->dst[c_] != s_->src[c_]) ZK_HIP(s_->ctx, hipMemcpyAsync(s_->dst[c_], s_->src[c_], s_->body, s_->kind, s_->ctx->stream_copy));      // in place: only the blinding rows move
            ZK_HIP(s_->ctx, hipMemcpyAsync((char*)s_->dst[c_] + s_->body, s_->blind + c_ * (s_->tail / 32), s_->tail, hipMemcpyHostToDevice, s_->ctx->stream_copy));
        }
        PK_TRY(copy_stream_fence(s_->ctx));
        ZK_HIP(s_->ctx, hipStreamWaitEvent(s_->ctx->stream_aux, s_->ctx->ev_copy, 0));      // the aux stream transforms what was just uploaded
        return ZK_OK;
    };
    std::vector<G1Affine> coms(sg.dst.size());
    std::vector<uint8_t> narrow(sg.src.size());
    if (owner_only) {                             // only this rank's own columns are on this device (and only they are committed here)
        std::vector<const void*> own_src;
        for (size_t c_ = pr->rank; c_ < sg.src.size(); c_ += pr->world) own_src.push_back(sg.src[c_]);
        std::vector<uint8_t> own_narrow(own_src.size());
        PK_TRY(sample_narrow_dev(ctx, own_src.data(), own_src.size(), n, own_narrow.data()));
        for (size_t j = 0; j < own_src.size(); ++j) narrow[pr->rank + j * pr->world] = own_narrow[j];
    } else if (dev_src) PK_TRY(sample_narrow_dev(ctx, sg.src.data(), sg.src.size(), n, narrow.data()));
    else sample_narrow(sg.src.data(), sg.src.size(), n, narrow.data());       // witness columns of (mostly) small values take the per-window MSM path
    trace.mark("  advice: columns sampled");
    // ... and their blinding rows (the last blinding_factors + 1, field-sized) are committed apart, so that they do not occupy every window
    struct TailGuard { zk_ctx* c; ~TailGuard() { c->msm_blinded_tail = 0; } } tail_guard{ctx};
    ctx->msm_blinded_tail = pk->bf + 1;
    if (sg.world == 1) {
        PK_TRY(commit_batch_staged(ctx, pk->srs, 1, (const void* const*)sg.dst.data(), sg.dst.size(), n, coms.data(), stage, &sg, narrow.data()));
        if (!sg.dst.empty()) {                    // the last column (its upload was fenced into the auxiliary stream by the last staging call) and what is pending
            sg.pending.push_back(sg.dst.size() - 1);
            PK_TRY(sg.flush());
        }
    } else {
        const size_t total = sg.dst.size(), per = (total + sg.world - 1) / sg.world;
        std::vector<const void*> mine;
        std::vector<uint8_t> mine_narrow;
        for (size_t c_ = pr->rank; c_ < total; c_ += sg.world) { mine.push_back(sg.dst[c_]); mine_narrow.push_back(narrow[c_]); }
        std::vector<G1Affine> local(per), all(per * sg.world);
        memset((void*)local.data(), 0, sizeof(G1Affine) * per);
        if (pr->gather_dev) {
            // Device all-gather mode: this rank uploads only ITS columns (1/world of the PCIe traffic),
            // commits them, and the ranks then exchange the columns group by group over the fabric.
            for (size_t c_ = pr->rank; c_ < total; c_ += sg.world) sg.own.push_back(c_);
            PK_TRY(commit_batch_staged(ctx, pk->srs, 1, mine.data(), mine.size(), n, local.data(), stage, &sg, mine_narrow.data()));
            // The exchange goes SEVERAL groups of `world` columns at a time (ZK_SHARD_EXCHANGE_GROUPS, default 16): this rank's columns of the chunk are packed into one
            // send buffer and ONE all-gather moves world x 16 columns -- fewer, larger collectives (a collective per column group was 125 host-synchronised rounds of
            // 32 MB per rank at eight ranks; RCCL over point-to-point xGMI wants its messages large).
            static const size_t xg_knob = getenv("ZK_SHARD_EXCHANGE_GROUPS") ? (size_t)atol(getenv("ZK_SHARD_EXCHANGE_GROUPS")) : 16;
            const size_t groups_total = (total + sg.world - 1) / sg.world, XG = std::max<size_t>(1, std::min(xg_knob, groups_total));
            DevBuf gbuf, sbuf;
            if (!gbuf.alloc((size_t)sg.world * XG * n * 32) || !sbuf.alloc(XG * n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
            // the last own column of the phase has not been transformed yet (the staging callback transforms column i - 1 when it uploads column i; the auxiliary
            // stream already waits for its upload): the exchange below may ship its coefficient form
            for (size_t c_ : sg.own) if (!sg.coeff[c_]->p) PK_TRY(to_coeff_aux(ctx, pk, *sg.lag[c_], sg.coeff[c_]));
            // the transforms of this rank's own columns still run on the auxiliary stream and share the NTT
            // scratch with the ones enqueued below on the main stream: finish them first
            ZK_HIP(ctx, hipStreamSynchronize(ctx->stream_aux));
            // WHAT travels (round 6): the owner has transformed its column already, and a column that no program over the Lagrange domain reads on this side of the
            // boundary -- no lookup tuple, not a permutation column, no remainder of the additive split: on the EVM-style shape 544 of 1000 -- is needed by its peers in
            // coefficient form only (cosets, evaluations, multi-open).  It is shipped in THAT form: same bytes, no inverse transform on the seven other ranks.
            // ZK_SHARD_COEFF=0 ships every column as Lagrange values (and every rank transforms every column), as before.
            std::vector<uint8_t> need_lag;
            PK_TRY(advice_lagrange_readers(ctx, pk, &need_lag));
            static const bool ship_coeff = !(getenv("ZK_SHARD_COEFF") && atoi(getenv("ZK_SHARD_COEFF")) == 0);
            auto as_coeff = [&](size_t c_) { return ship_coeff && !need_lag[sg.col[c_]]; };
            for (size_t grp0 = 0; grp0 < groups_total; grp0 += XG) {
                const size_t gcnt = std::min(XG, groups_total - grp0);
                for (size_t g = 0; g < gcnt; ++g) {
                    const size_t mine_c = (grp0 + g) * sg.world + pr->rank;
                    if (mine_c < total) {
                        const void* src_ = sg.dst[mine_c];
                        if (as_coeff(mine_c)) {
                            if (!sg.coeff[mine_c]->p) return ctx->fail(ZK_ERR_INVALID_ARG, "sharded session: the coefficient form of an own column is missing at the exchange");
                            src_ = sg.coeff[mine_c]->p;
                        }
                        ZK_HIP(ctx, hipMemcpyAsync((char*)sbuf.p + g * n * 32, src_, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
                    } else ZK_HIP(ctx, hipMemsetAsync((char*)sbuf.p + g * n * 32, 0, n * 32, ctx->stream));
                }
                PK_TRY(zk_ctx_sync(ctx));                                  // own columns uploaded and packed, previous copies out of gbuf done
                if (pr->gather_dev(pr->gather_dev_user, sbuf.p, gcnt * n * 32, gbuf.p))
                    return ctx->fail(ZK_ERR_INVALID_ARG, "sharded session: device all-gather callback failed");
                for (size_t g = 0; g < gcnt; ++g)
                    for (uint32_t q_ = 0; q_ < sg.world; ++q_) {
                        const size_t c_ = (grp0 + g) * sg.world + q_;
                        if (q_ == pr->rank || c_ >= total) continue;
                        const char* got = (const char*)gbuf.p + ((size_t)q_ * gcnt + g) * n * 32;
                        if (as_coeff(c_)) {
                            if (!sg.coeff[c_]->alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
                            ZK_HIP(ctx, hipMemcpyAsync(sg.coeff[c_]->p, got, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
                            sg.lag[c_]->release();                   // nobody reads the Lagrange form of this column here
                            pr->lag_partial = true;
                        } else {
                            ZK_HIP(ctx, hipMemcpyAsync(sg.dst[c_], got, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
                            PK_TRY(to_coeff(ctx, pk, *sg.lag[c_], sg.coeff[c_]));
                        }
                    }
            }
        } else {
            PK_TRY(commit_batch_staged(ctx, pk->srs, 1, mine.data(), mine.size(), n, local.data(), stage, &sg, mine_narrow.data()));   // MSM j reads group j of `world` columns
            for (size_t grp = mine.size(); grp * sg.world < total; ++grp) PK_TRY(stage(&sg, grp));                   // a last group without a column of this rank
        }
        PK_TRY(zk_ctx_sync(ctx));
        if (per && pr->gather(pr->gather_user, local.data(), per * sizeof(G1Affine), all.data())) return ctx->fail(ZK_ERR_INVALID_ARG, "sharded session: all-gather callback failed");
        for (size_t c_ = 0; c_ < total; ++c_) coms[c_] = all[(c_ % sg.world) * per + c_ / sg.world];
    }
    ZK_HIP(ctx, hipEventRecord(ctx->ev_aux, ctx->stream_aux));             // the main stream continues after the transforms
    ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_aux, 0));
    trace.mark("advice upload + commits");
    for (const G1Affine& com : coms) pr->tr.write_point(com);
    uint32_t written = 0;
    for (uint32_t i = 0; i < pk->chal_phase.size(); ++i) {
        if (pk->chal_phase[i] != pr->phase) continue;
        pr->challenges[i] = pr->tr.squeeze();
        if (h_challenges) memcpy((uint8_t*)h_challenges + 32 * written, &pr->challenges[i], 32);
        ++written;
    }
    if (num_challenges) *num_challenges = written;
    if (pr->tr.err) return ctx->fail(ZK_ERR_INVALID_ARG, "external transcript callback failed with status %d", pr->tr.err);
    pr->absorbed.clear();
    pr->absorbed.shrink_to_fit();
    ++pr->phase;
    return ZK_OK;
}

// The quotient's constraints in halo2's order: gates, permutation, lookups (folded with y by the caller).  A pure function of
// the key: challenges enter as constant indices, so the programs -- and with them which column is read on which coset -- are
// known before any witness exists (the advice phase uses that to transform columns ahead of time, advice_coset_plan).
static void build_constraints(const zk_pk* pk, std::vector<Prog>& cons, bool& gates_share_tmps_out) {
    PB q;
    auto end_c = [&] { cons.push_back(std::move(q.g)); q.g.clear(); };
    for (const Prog& g : pk->gates) cons.push_back(g);
    const int32_t rot_last = -(int32_t)(pk->bf + 1);
    if (pk->C) {
        q.col(CT_SPECIAL, SP_L0).cst(C_ONE).col(CT_PERM_Z, 0).op(Q_SUB).op(Q_MUL); end_c();                                   // l0 (1 - Z_0)
        q.col(CT_SPECIAL, SP_LLAST).col(CT_PERM_Z, pk->C - 1).op(Q_SQUARE).col(CT_PERM_Z, pk->C - 1).op(Q_SUB).op(Q_MUL); end_c();   // l_last (Z^2 - Z)
        for (uint32_t c = 1; c < pk->C; ++c) {
            q.col(CT_SPECIAL, SP_L0).col(CT_PERM_Z, c).col(CT_PERM_Z, c - 1, rot_last).op(Q_SUB).op(Q_MUL);          // l0 (Z_c - Z_{c-1}(w^last X))
            end_c();
        }
        for (uint32_t c = 0; c < pk->C; ++c) {
            const uint32_t j0 = c * pk->chunk, j1 = std::min(pk->P, j0 + pk->chunk);
            q.col(CT_SPECIAL, SP_LACTIVE);
            q.col(CT_PERM_Z, c, 1);
            for (uint32_t j = j0; j < j1; ++j) { push_perm_col(q, pk->perm_cols[j]); q.col(CT_SIGMA, j).mulc(C_BETA).op(Q_ADD).addc(C_GAMMA).op(Q_MUL); }
            q.col(CT_PERM_Z, c, 0);
            for (uint32_t j = j0; j < j1; ++j) { push_perm_col(q, pk->perm_cols[j]); q.col(CT_SPECIAL, SP_X).mulc(C_DELTA0 + j).op(Q_ADD).addc(C_GAMMA).op(Q_MUL); }
            q.op(Q_SUB).op(Q_MUL); end_c();
        }
    }
    // lookup identities (plonk::evaluation, mv-lookup): with phi_a = f_a + beta and tau = t + beta,
    //   l_active * ( tau * prod_a phi_a * (phi(wX) - phi(X))  -  prod_a phi_a * (tau * sum_a 1/phi_a - m) )
    // An argument with several input tuples parks tau and the phi_a of a row in intermediates (slots
    // above the ones the gate programs use) and forms sum_a prod_{b != a} phi_b from them.
    uint32_t tmp_base = 0;
    for (const Prog& g : pk->gates) for (const Instr& in : g) if (in.op == Q_TEE_TMP || in.op == Q_PUSH_TMP) tmp_base = std::max(tmp_base, in.a + 1);
    const bool gates_share_tmps = tmp_base != 0;
    // Single-tuple lookups share the table side: tau = t + beta of a table is parked by the first lookup into it and read back by the
    // others (slots above the ones the multi-tuple form reuses; defined once each, so degree classes re-materialise them, TmpSplit).
    uint32_t max_tuples = 0;
    for (uint32_t l = 0; l < pk->L; ++l) if (pk->lookups[l].inputs.size() > 1) max_tuples = std::max<uint32_t>(max_tuples, (uint32_t)pk->lookups[l].inputs.size());
    const uint32_t table_slot0 = tmp_base + (max_tuples ? max_tuples + 2 : 0);
    std::vector<const std::vector<Prog>*> parked_tables;               // table of slot table_slot0 + i
    static const bool lk_plain = getenv("ZK_LOOKUP_PLAIN") && atoi(getenv("ZK_LOOKUP_PLAIN")) == 1;      // measurement knob: the identity as halo2 writes it
    auto same_tables = [](const std::vector<Prog>& x, const std::vector<Prog>& y) {
        if (x.size() != y.size()) return false;
        for (size_t i = 0; i < x.size(); ++i) {
            if (x[i].size() != y[i].size()) return false;
            for (size_t j = 0; j < x[i].size(); ++j) if (x[i][j].op != y[i][j].op || x[i][j].a != y[i][j].a || x[i][j].b != y[i][j].b) return false;
        }
        return true;
    };
    for (uint32_t l = 0; l < pk->L; ++l) {
        const auto& lk = pk->lookups[l];
        const uint32_t N = (uint32_t)lk.inputs.size();
        q.col(CT_SPECIAL, SP_L0).col(CT_LK_PHI, l).op(Q_MUL); end_c();
        q.col(CT_SPECIAL, SP_LLAST).col(CT_LK_PHI, l).op(Q_MUL); end_c();
        q.col(CT_SPECIAL, SP_LACTIVE);
        if (N == 1 && lk_plain) {
            // (phi(wX) - phi(X)) (f+beta)(t+beta) - ((t+beta) - m (f+beta))
            q.col(CT_LK_PHI, l, 1).col(CT_LK_PHI, l, 0).op(Q_SUB);
            push_compressed(q, lk.inputs[0]); q.addc(C_BETA).op(Q_MUL);
            push_compressed(q, lk.tables); q.addc(C_BETA).op(Q_MUL);
            push_compressed(q, lk.tables); q.addc(C_BETA);
            q.col(CT_LK_M, l); push_compressed(q, lk.inputs[0]); q.addc(C_BETA).op(Q_MUL);
            q.op(Q_SUB).op(Q_SUB).op(Q_MUL); end_c();
            continue;
        }
        if (N == 1) {
            // the same polynomial with phi_f = f + beta and tau = t + beta each computed once:
            //   (phi(wX) - phi(X)) phi_f tau - (tau - m phi_f)  =  phi_f ((phi(wX) - phi(X)) tau + m) - tau
            // four products instead of twelve on a two-column tuple under one selector (push_compressed takes the selector out)
            size_t ti = 0;
            while (ti < parked_tables.size() && !same_tables(*parked_tables[ti], lk.tables)) ++ti;
            const bool first_use = ti == parked_tables.size();
            if (first_use) parked_tables.push_back(&lk.tables);
            const uint32_t slot = table_slot0 + (uint32_t)ti;
            q.col(CT_LK_PHI, l, 1).col(CT_LK_PHI, l, 0).op(Q_SUB);
            if (first_use) { push_compressed(q, lk.tables); q.addc(C_BETA); q.g.push_back({Q_TEE_TMP, slot, 0}); }
            else q.g.push_back({Q_PUSH_TMP, slot, 0});
            q.op(Q_MUL);
            q.col(CT_LK_M, l).op(Q_ADD);
            push_compressed(q, lk.inputs[0]); q.addc(C_BETA).op(Q_MUL);
            q.g.push_back({Q_PUSH_TMP, slot, 0});
            q.op(Q_SUB).op(Q_MUL); end_c();
            continue;
        }
        const uint32_t t_tau = tmp_base, t_phi = tmp_base + 1;          // reused by every multi-input lookup: a row's values are consumed right away
        auto tmp = [&](uint32_t op, uint32_t slot) { q.g.push_back({op, slot, 0}); };
        // prod = phi_0 * ... * phi_{N-1}, each factor parked on the way
        for (uint32_t a = 0; a < N; ++a) {
            push_compressed(q, lk.inputs[a]); q.addc(C_BETA);
            tmp(Q_TEE_TMP, t_phi + a);
            if (a) q.op(Q_MUL);
        }
        tmp(Q_TEE_TMP, t_phi + N);                                       // prod
        // lhs = tau * prod * (phi(wX) - phi(X))
        push_compressed(q, lk.tables); q.addc(C_BETA);
        tmp(Q_TEE_TMP, t_tau);
        q.op(Q_MUL);
        q.col(CT_LK_PHI, l, 1).col(CT_LK_PHI, l, 0).op(Q_SUB).op(Q_MUL);
        // rhs = tau * sum_a prod_{b != a} phi_b - prod * m
        for (uint32_t a = 0; a < N; ++a) {
            bool first = true;
            for (uint32_t b2 = 0; b2 < N; ++b2) {
                if (b2 == a) continue;
                tmp(Q_PUSH_TMP, t_phi + b2);
                if (!first) q.op(Q_MUL);
                first = false;
            }
            if (a) q.op(Q_ADD);
        }
        tmp(Q_PUSH_TMP, t_tau); q.op(Q_MUL);
        tmp(Q_PUSH_TMP, t_phi + N); q.col(CT_LK_M, l).op(Q_MUL);
        q.op(Q_SUB);
        q.op(Q_SUB).op(Q_MUL); end_c();
    }
    gates_share_tmps_out = gates_share_tmps;
}
// e = ceil(log2(degree - 1)) of every constraint (its degree class, see zk_proof_finish), or E for all when classes are off
static int classify_constraints(std::string* err, const zk_pk* pk, const std::vector<Prog>& cons, bool split, uint32_t E, std::vector<uint32_t>& cls) {
    cls.assign(cons.size(), E);
    std::vector<int> tmp_deg;
    char msg[160];
    for (uint32_t i = 0; i < cons.size(); ++i) {
        const int dg = program_degree(cons[i], &tmp_deg);
        if (dg < 0) { snprintf(msg, sizeof msg, "prover: malformed constraint program %u", i); *err = msg; return ZK_ERR_INVALID_ARG; }
        if ((uint32_t)dg > pk->d) { snprintf(msg, sizeof msg, "prover: constraint %u has degree %d above the circuit degree %u", i, dg, pk->d); *err = msg; return ZK_ERR_INVALID_ARG; }
        uint32_t e = 0;
        while (e < E && ((uint32_t)1 << e) < (uint32_t)std::max(dg - 1, 1)) ++e;
        cls[i] = split ? e : E;
    }
    return ZK_OK;
}
// Additive split of a constraint over degree classes.  h is linear in the constraints, so a constraint that is a SUM of terms of
// different degrees -- q (a b - c): the product is of degree 3, q c of degree 2 -- may put each term into the class of ITS degree,
// all with the constraint's own power of y: the term q c is then evaluated on one coset instead of two, and a column that occurs
// only in such low-degree terms (the outputs of multiplication gates, every linearly constrained cell) is transformed to fewer
// cosets.  Recognised shapes (the program's top): SUM, F * SUM and SUM * F with SUM a tree of + / - / negations; the factor F (a
// selector, as a rule) multiplies every group.  The groups sum to the constraint, but only the constraint vanishes on H: what a group
// that moves to a lower class is on H travels with it as a remainder polynomial (zk_proof_finish, CT_SPLIT_R), so that every class
// still divides by X^n - 1 exactly.  Exact field arithmetic: h and every proof byte stay what they were.  Programs that park or
// read shared intermediates are left whole.  ZK_QUOTIENT_ADDSPLIT=0 turns it off.
struct ClassPiece { uint32_t cons, cls; Prog prog; };
static int TmpSplitArity(uint32_t op) {
    switch (op) {
        case Q_PUSH_COL: case Q_PUSH_CONST: case Q_PUSH_TMP: return 0;
        case Q_ADD: case Q_SUB: case Q_MUL: return 2;
        default: return 1;                        // NEG, DOUBLE, SQUARE, ADD_CONST, MUL_CONST, TEE_TMP
    }
}
static uint32_t class_of_degree(int dg, uint32_t E) {
    uint32_t e = 0;
    while (e < E && ((uint32_t)1 << e) < (uint32_t)std::max(dg - 1, 1)) ++e;
    return e;
}
static bool additive_split(const Prog& g, uint32_t E, uint32_t whole_cls, std::vector<std::pair<uint32_t, Prog>>& out) {
    struct Node { Instr in; int l, r; };
    std::vector<Node> nd;
    std::vector<int> st;
    for (const Instr& in : g) {
        if (in.op == Q_TEE_TMP || in.op == Q_PUSH_TMP || in.op == Q_FOLD || in.op == Q_END) return false;
        const int ar = TmpSplitArity(in.op);
        Node n_{in, -1, -1};
        if (ar == 2) { if (st.size() < 2) return false; n_.r = st.back(); st.pop_back(); n_.l = st.back(); st.pop_back(); }
        else if (ar == 1) { if (st.empty()) return false; n_.l = st.back(); st.pop_back(); }
        nd.push_back(n_);
        st.push_back((int)nd.size() - 1);
    }
    if (st.size() != 1) return false;
    const int root = st[0];
    std::function<void(int, Prog&)> emit = [&](int v, Prog& o) {
        if (nd[v].l >= 0) emit(nd[v].l, o);
        if (nd[v].r >= 0) emit(nd[v].r, o);
        o.push_back(nd[v].in);
    };
    auto is_sum = [&](int v) { return nd[v].in.op == Q_ADD || nd[v].in.op == Q_SUB || nd[v].in.op == Q_NEG; };
    int factor = -1, sum = root;
    if (nd[root].in.op == Q_MUL) {
        if (is_sum(nd[root].r)) { factor = nd[root].l; sum = nd[root].r; }
        else if (is_sum(nd[root].l)) { factor = nd[root].r; sum = nd[root].l; }
        else return false;
    } else if (!is_sum(root)) return false;
    std::vector<std::pair<int, bool>> terms;            // (node, negative)
    std::function<void(int, bool)> collect = [&](int v, bool neg_) {
        const uint32_t op = nd[v].in.op;
        if (op == Q_ADD) { collect(nd[v].l, neg_); collect(nd[v].r, neg_); }
        else if (op == Q_SUB) { collect(nd[v].l, neg_); collect(nd[v].r, !neg_); }
        else if (op == Q_NEG) collect(nd[v].l, !neg_);
        else terms.push_back({v, neg_});
    };
    collect(sum, false);
    if (terms.size() < 2 || terms.size() > 4096) return false;
    std::vector<int> no_tmps;
    Prog fprog;
    int fdeg = 0;
    if (factor >= 0) { emit(factor, fprog); fdeg = program_degree(fprog, &no_tmps); if (fdeg < 0) return false; }
    std::vector<Prog> tprog(terms.size());
    std::vector<uint32_t> tcls(terms.size());
    for (size_t t = 0; t < terms.size(); ++t) {
        emit(terms[t].first, tprog[t]);
        const int tdeg = program_degree(tprog[t], &no_tmps);
        if (tdeg < 0) return false;
        tcls[t] = class_of_degree(fdeg + tdeg, E);
    }
    std::vector<Prog> group(E + 1);
    std::vector<uint8_t> started(E + 1, 0);
    for (uint32_t e = 0; e <= E; ++e) {
        int lead = -1;                                   // a positive term first (no negation to spend), else the first term negated
        for (size_t t = 0; t < terms.size() && lead < 0; ++t) if (tcls[t] == e && !terms[t].second) lead = (int)t;
        for (size_t t = 0; t < terms.size() && lead < 0; ++t) if (tcls[t] == e) lead = (int)t;
        if (lead < 0) continue;
        started[e] = 1;
        group[e] = tprog[lead];
        if (terms[lead].second) group[e].push_back({Q_NEG, 0, 0});
        for (size_t t = 0; t < terms.size(); ++t) {
            if (tcls[t] != e || (int)t == lead) continue;
            group[e].insert(group[e].end(), tprog[t].begin(), tprog[t].end());
            group[e].push_back({terms[t].second ? Q_SUB : Q_ADD, 0, 0});
        }
    }
    uint32_t used = 0;
    for (uint32_t e = 0; e <= E; ++e) used += started[e] != 0;
    if (used < 2) return false;
    (void)whole_cls;
    for (uint32_t e = 0; e <= E; ++e) {
        if (!started[e]) continue;
        Prog pg = fprog;
        pg.insert(pg.end(), group[e].begin(), group[e].end());
        if (factor >= 0) pg.push_back({Q_MUL, 0, 0});
        out.push_back({e, std::move(pg)});
    }
    return true;
}
// the pieces the classes evaluate, in constraint order (a constraint's pieces by ascending class)
static bool quotient_addsplit_enabled() {
    const char* env = getenv("ZK_QUOTIENT_ADDSPLIT");
    return !(env && atoi(env) == 0);
}
static void class_pieces(const std::vector<Prog>& cons, const std::vector<uint32_t>& cls, bool split, bool addsplit, uint32_t E, std::vector<ClassPiece>& out) {
    const bool on = split && addsplit;
    for (uint32_t i = 0; i < cons.size(); ++i) {
        std::vector<std::pair<uint32_t, Prog>> parts;
        if (on && cls[i] > 0 && additive_split(cons[i], E, cls[i], parts)) {
            for (auto& pt : parts) out.push_back({i, pt.first, std::move(pt.second)});
        } else out.push_back({i, cls[i], cons[i]});
    }
}
// A class program as ONE weighted sum (round 5).  The class's accumulator used to fold its terms one by one, acc = acc * y^gap + term:
// a product per term, and every term carried its selector -- q (a b), q' c, l_active (...) -- as a product of its own.  The sum
//      sum_i y^(K-1-i) t_i      (t_i = the terms of this class, i = the constraint a term belongs to)
// is the same polynomial when the terms that share a single-column factor F are collected first:
//      sum_F F * (sum_{i in F} y^(K-1-i) r_i) + sum_{others} y^(K-1-i) t_i,        t_i = F r_i
// -- one product by F per group instead of one per term, and no product for the folding.  Exact field arithmetic: h and every proof
// byte are unchanged (tests/test_quotient_classes.py evaluates both forms; the GPU proof tests compare bytes with the oracle prover).
// A term may use parked intermediates: it joins a group only if everything it reads was parked before the group's first term, and a
// term that parks something only ever opens a group, so definitions stay ahead of their readers when later terms move up.
// ZK_QUOTIENT_GROUP=0 keeps the folded form.
struct ClassTerm { uint32_t cons; Prog prog; };
static bool assemble_grouped(const std::vector<ClassTerm>& terms, uint32_t K, Prog& out) {
    if (terms.empty() || K >= 0xFFFFu) return false;
    struct Info { bool has_factor[2] = {false, false}; Instr factor[2]; Prog rest[2]; std::vector<uint32_t> reads, defs; int pick = -1; };
    std::vector<Info> info(terms.size());
    auto key = [](const Instr& f) { return ((uint64_t)f.a << 32) | f.b; };
    std::unordered_map<uint64_t, uint32_t> count;
    for (size_t t = 0; t < terms.size(); ++t) {
        const Prog& g = terms[t].prog;
        for (const Instr& in : g) {
            if (in.op == Q_PUSH_TMP) info[t].reads.push_back(in.a);
            else if (in.op == Q_TEE_TMP) info[t].defs.push_back(in.a);
            else if (in.op == Q_FOLD || in.op == Q_END) return false;
        }
        for (int first = 0; first < 2; ++first) {
            info[t].has_factor[first] = split_leaf_factor(g, first != 0, &info[t].rest[first], &info[t].factor[first]);
            if (info[t].has_factor[first] && !(first == 1 && info[t].has_factor[0] && key(info[t].factor[0]) == key(info[t].factor[1]))) ++count[key(info[t].factor[first])];
        }
    }
    for (size_t t = 0; t < terms.size(); ++t) {
        uint32_t best = 1;          // a factor no other term shares is left where it is
        for (int first = 1; first >= 0; --first)
            if (info[t].has_factor[first] && count[key(info[t].factor[first])] > best) { best = count[key(info[t].factor[first])]; info[t].pick = first; }
    }
    // groups in the order of their first terms
    struct Group { std::vector<uint32_t> members; bool factored; std::vector<uint32_t> known; };       // known: what was parked before (and by) the first term
    std::vector<Group> groups;
    std::unordered_map<uint64_t, uint32_t> open;
    std::vector<uint32_t> parked;                      // slots parked by the terms seen so far (original order)
    std::unordered_map<uint32_t, uint32_t> ndefs;      // a slot that is parked more than once (reused) is only read where it stands
    for (const Info& ti : info) for (uint32_t d : ti.defs) ++ndefs[d];
    auto stable = [&](const std::vector<uint32_t>& reads) { for (uint32_t v : reads) { auto it = ndefs.find(v); if (it == ndefs.end() || it->second != 1) return false; } return true; };
    auto subset = [](const std::vector<uint32_t>& x, const std::vector<uint32_t>& of) { for (uint32_t v : x) if (std::find(of.begin(), of.end(), v) == of.end()) return false; return true; };
    for (size_t t = 0; t < terms.size(); ++t) {
        const Info& ti = info[t];
        bool joined = false;
        if (ti.pick >= 0 && ti.defs.empty() && stable(ti.reads)) {
            auto it = open.find(key(ti.factor[ti.pick]));
            if (it != open.end() && subset(ti.reads, groups[it->second].known)) { groups[it->second].members.push_back((uint32_t)t); joined = true; }
        }
        if (!joined) {
            Group gnew;
            gnew.members.push_back((uint32_t)t);
            gnew.factored = ti.pick >= 0;
            gnew.known = parked;
            // what the first term parks inside its OWN co-factor is computed before any other member's co-factor is
            if (ti.pick >= 0) for (const Instr& in : ti.rest[ti.pick]) if (in.op == Q_TEE_TMP) gnew.known.push_back(in.a);
            groups.push_back(std::move(gnew));
            if (ti.pick >= 0) open[key(ti.factor[ti.pick])] = (uint32_t)groups.size() - 1;
        }
        for (uint32_t d : ti.defs) parked.push_back(d);
    }
    out.clear();
    auto weight = [&](uint32_t cons) { if (K - 1 - cons) out.push_back({Q_MUL_CONST, C_YPOW0 + (K - 1 - cons), 0}); };
    bool first_item = true;
    for (const Group& gr : groups) {
        if (gr.factored && gr.members.size() >= 2) {
            const Info& lead = info[gr.members[0]];
            for (size_t j = 0; j < gr.members.size(); ++j) {
                const uint32_t t = gr.members[j];
                const Prog& r = info[t].rest[info[t].pick];
                out.insert(out.end(), r.begin(), r.end());
                weight(terms[t].cons);
                if (j) out.push_back({Q_ADD, 0, 0});
            }
            out.push_back(lead.factor[lead.pick]);
            out.push_back({Q_MUL, 0, 0});
        } else {
            for (size_t j = 0; j < gr.members.size(); ++j) {          // a lone term (a group nobody joined)
                const uint32_t t = gr.members[j];
                out.insert(out.end(), terms[t].prog.begin(), terms[t].prog.end());
                weight(terms[t].cons);
                if (j) out.push_back({Q_ADD, 0, 0});
            }
        }
        if (!first_item) out.push_back({Q_ADD, 0, 0});
        first_item = false;
    }
    out.push_back({Q_FOLD, C_ONE, 0});                  // acc = 0 * 1 + the sum
    // depth of the caller's stack machine (Q_MAX_STACK in quotient.hip is 16): beyond it the folded form is kept
    int sp = 0, mx = 0;
    for (const Instr& in : out) {
        if (in.op == Q_PUSH_COL || in.op == Q_PUSH_CONST || in.op == Q_PUSH_TMP) ++sp;
        else if (in.op == Q_ADD || in.op == Q_SUB || in.op == Q_MUL || in.op == Q_FOLD) --sp;
        mx = std::max(mx, sp);
    }
    return sp == 0 && mx <= 14;
}
static bool quotient_group_enabled() {
    const char* env = getenv("ZK_QUOTIENT_GROUP");
    return !(env && atoi(env) == 0);
}
#include "class_compile.hpp"
static bool quotient_dag_enabled() {
    const char* env = getenv("ZK_QUOTIENT_DAG");
    return !(env && atoi(env) == 0);
}
// Intermediates shared between constraints (TEE_TMP in one gate, PUSH_TMP in a later one: the common-subexpression
// elimination of halo2's GraphEvaluator as it survives the export) and degree classes: a class evaluates only ITS constraints,
// so a class that reads an intermediate another class parked must compute it itself.  `TmpSplit` re-materialises: walking the
// constraints in order, a PUSH_TMP whose definition this class has not emitted yet is replaced by the defining sub-expression
// (its own PUSH_TMPs resolved the same way) followed by the TEE_TMP; afterwards the class reads the slot like any other.
// Classes run as separate launches over the same parking area, each defining what it reads before reading it.
// Not handled (ok() == false, the caller evaluates everything as one class, as before): a slot that is defined more than once
// AND read by a constraint other than the one that defined it (slot reuse with cross-constraint lifetime).
struct TmpSplit {
    std::vector<Prog> defs;                       // slot -> defining sub-expression (postfix, without the TEE)
    std::vector<uint32_t> ver;                    // slot -> number of definitions seen so far
    std::vector<std::vector<uint32_t>> have;      // class -> slot -> version this class has emitted (0 = none)
    bool conflict = false;
    static int arity(uint32_t op) {
        switch (op) {
            case Q_PUSH_COL: case Q_PUSH_CONST: case Q_PUSH_TMP: return 0;
            case Q_ADD: case Q_SUB: case Q_MUL: return 2;
            default: return 1;                    // NEG, DOUBLE, SQUARE, ADD_CONST, MUL_CONST, TEE_TMP (peeks: one in, one out)
        }
    }
    // the sub-expression whose value is on top of the stack after instruction t - 1 of g
    static Prog sub_expression(const Prog& g, size_t t) {
        int need = 1;
        size_t start = t;
        while (start > 0 && need > 0) { --start; need += arity(g[start].op) - 1; }
        return Prog(g.begin() + start, g.begin() + t);
    }
    explicit TmpSplit(size_t classes) : have(classes) {}
    void grow(uint32_t s) {
        if (s >= ver.size()) { ver.resize(s + 1, 0); defs.resize(s + 1); }
        for (auto& h : have) if (s >= h.size()) h.resize(s + 1, 0);
    }
    void emit_push(uint32_t s, uint32_t e, Prog& out, int depth = 0) {
        grow(s);
        if (have[e][s] == ver[s] && ver[s]) { out.push_back({Q_PUSH_TMP, s, 0}); return; }
        if (!ver[s] || depth > 64) { conflict = true; out.push_back({Q_PUSH_TMP, s, 0}); return; }     // read before any definition: malformed
        for (const Instr& in : defs[s]) {
            if (in.op == Q_PUSH_TMP) emit_push(in.a, e, out, depth + 1);
            else {
                if (in.op == Q_TEE_TMP) { grow(in.a); have[e][in.a] = ver[in.a]; }
                out.push_back(in);
            }
        }
        out.push_back({Q_TEE_TMP, s, 0});
        have[e][s] = ver[s];
    }
    // appends constraint g (class e) to out with its intermediates resolved for that class
    void append(const Prog& g, uint32_t e, Prog& out) {
        for (size_t t = 0; t < g.size(); ++t) {
            const Instr& in = g[t];
            if (in.op == Q_TEE_TMP) {
                grow(in.a);
                defs[in.a] = sub_expression(g, t);
                ++ver[in.a];
                have[e][in.a] = ver[in.a];
                out.push_back(in);
            } else if (in.op == Q_PUSH_TMP) emit_push(in.a, e, out);
            else out.push_back(in);
        }
    }
};
// slot reuse with cross-constraint lifetime (see TmpSplit): such keys keep the single-class evaluation
static bool tmp_slots_conflict(const std::vector<Prog>& cons) {
    std::vector<uint32_t> ndef, def_at;
    std::vector<uint8_t> cross;
    auto grow = [&](uint32_t s) { if (s >= ndef.size()) { ndef.resize(s + 1, 0); def_at.resize(s + 1, 0); cross.resize(s + 1, 0); } };
    for (uint32_t i = 0; i < cons.size(); ++i)
        for (const Instr& in : cons[i]) {
            if (in.op == Q_TEE_TMP) { grow(in.a); ++ndef[in.a]; def_at[in.a] = i; }
            else if (in.op == Q_PUSH_TMP) { grow(in.a); if (!ndef[in.a]) return true; if (def_at[in.a] != i) cross[in.a] = 1; }
        }
    for (size_t s_ = 0; s_ < ndef.size(); ++s_) if (ndef[s_] > 1 && cross[s_]) return true;
    return false;
}
static bool quotient_split_enabled(bool sharded, bool gates_share_tmps) {
    const char* split_env = getenv("ZK_QUOTIENT_SPLIT");
    (void)gates_share_tmps;                   // shared intermediates are re-materialised per class (TmpSplit); slot-reuse conflicts are checked by the caller
    (void)sharded;                            // sharded sessions distribute (class, coset) pairs over the ranks (zk_proof_finish)
    return !(split_env && atoi(split_env) == 0);
}
// ---- the quotient's plan: which constraint (or part of one) is evaluated in which degree class, and each class's program --------
// A pure function of the key and of the measurement knobs, so it is made once per key (zk_pk::qplan) and shared by the advice
// phases (which cosets read which column) and zk_proof_finish.  Candidates: degree classes with the additive split, degree
// classes alone, one class -- the cheapest by a count of what each would execute (below) is taken; ZK_QUOTIENT_COSTGATE=0 takes
// the most split one the knobs allow, as rounds 3-5 did.
struct QPlanClass { Prog prog; std::vector<uint32_t> refs; uint32_t last = 0; bool used = false; uint32_t products = 0, parked = 0, max_live = 0, groups = 0; };
struct QPlanRem { uint32_t t = 0, e = 0; Prog prog; uint32_t last = 0; bool used = false; uint32_t products = 0; };
struct QuotientPlan {
    std::string key;                 // the knob values the plan was made under
    uint32_t K = 0, E = 0;
    bool split = false, addsplit = false, dag = false;
    std::vector<QPlanClass> cls;     // E + 1 classes
    std::vector<QPlanRem> rems;      // remainder polynomials of the additive split: evaluated per proof over the Lagrange forms
    std::vector<uint32_t> refs;      // every column any class reads
    double cost = 0;                 // the estimate the candidates were compared by (units: one product over n rows)
};
static uint32_t count_products(const Prog& g) {
    uint32_t c = 0;
    for (const Instr& in : g) c += in.op == Q_MUL || in.op == Q_SQUARE || in.op == Q_MUL_CONST || in.op == Q_FOLD;
    return c;
}
static int make_quotient_plan(const zk_pk* pk, bool split_req, bool addsplit_req, QuotientPlan& qp, std::string* err) {
    std::vector<Prog> cons;
    bool gates_share_tmps = false;
    build_constraints(pk, cons, gates_share_tmps);
    const uint32_t E = pk->ext_k - pk->k, K = (uint32_t)cons.size();
    const bool conflict = tmp_slots_conflict(cons);
    const bool split = split_req && !conflict;
    qp.K = K; qp.E = E; qp.split = split; qp.addsplit = split && addsplit_req; qp.dag = quotient_dag_enabled();
    qp.cls.assign(E + 1, QPlanClass());
    qp.rems.clear();
    qp.refs.clear();
    std::vector<uint32_t> cls;
    PK_TRY(classify_constraints(err, pk, cons, split, E, cls));
    TmpSplit tmps(E + 1);
    std::vector<ClassPiece> cpieces;
    class_pieces(cons, cls, split, qp.addsplit, E, cpieces);
    // A constraint vanishes on H; a PART of it does not, and only multiples of X^n - 1 may be divided class by class.  So the
    // parts that leave their constraint's class t for a lower class e take their values on H along: R_(t,e) = the polynomial of
    // degree < n that agrees on H with the y-weighted sum of those parts (evaluated over the Lagrange forms, one pass, one
    // inverse transform).  Class e evaluates (its parts - R), class t (its parts + R): both vanish on H again, the total is
    // unchanged.  R is read like a column (CT_SPLIT_R) on the cosets of the two classes.
    std::vector<std::vector<ClassTerm>> cterms(E + 1);       // the terms of every class in constraint order: (constraint, program)
    {
        std::vector<uint32_t> top(K, 0);
        for (const ClassPiece& pc : cpieces) top[pc.cons] = std::max(top[pc.cons], pc.cls);
        for (const ClassPiece& pc : cpieces) {
            const uint32_t i = pc.cons;
            cterms[pc.cls].emplace_back();
            cterms[pc.cls].back().cons = i;
            tmps.append(pc.prog, pc.cls, cterms[pc.cls].back().prog);                   // the constraint (or its terms of this class), shared intermediates resolved for this class
            if (pc.cls < top[i]) {
                size_t at = 0;
                while (at < qp.rems.size() && !(qp.rems[at].t == top[i] && qp.rems[at].e == pc.cls)) ++at;
                if (at == qp.rems.size()) { qp.rems.emplace_back(); qp.rems.back().t = top[i]; qp.rems.back().e = pc.cls; }
                QPlanRem& rm = qp.rems[at];
                rm.prog.insert(rm.prog.end(), pc.prog.begin(), pc.prog.end());
                rm.prog.push_back({Q_FOLD, rm.used ? C_YPOW0 + (i - rm.last) : C_Y, 0});
                rm.last = i;
                rm.used = true;
            }
        }
    }
    if (tmps.conflict) { *err = "prover: a constraint reads an intermediate before any constraint parked it"; return ZK_ERR_INVALID_ARG; }
    for (size_t j = 0; j < qp.rems.size(); ++j) {
        // both classes take R in as one more term at the very end of the constraint list (weight y^0)
        const uint32_t ref = colref(CT_SPLIT_R, (uint32_t)j);
        cterms[qp.rems[j].e].push_back({K - 1, Prog{{Q_PUSH_COL, ref, 0}, {Q_NEG, 0, 0}}});
        cterms[qp.rems[j].t].push_back({K - 1, Prog{{Q_PUSH_COL, ref, 0}}});
        qp.rems[j].products = count_products(qp.rems[j].prog);
    }
    // the class programs: compiled through the expression graph (class_compile.hpp); knob off or a program too deep for the evaluator's
    // stack: one weighted sum (assemble_grouped, round 5) or the terms folded one by one, acc = acc * y^gap + term
    const bool grouped = quotient_group_enabled() && !conflict;
    for (uint32_t e = 0; e <= E; ++e) {
        QPlanClass& c = qp.cls[e];
        if (cterms[e].empty()) continue;
        c.used = true;
        ClassCompileStats st;
        bool done = qp.dag && compile_class(cterms[e], K, c.prog, &c.last, &st);
        if (done) { c.parked = st.parked; c.max_live = st.max_live; c.groups = st.groups; }
        if (!done && grouped && assemble_grouped(cterms[e], K, c.prog)) { c.last = K - 1; done = true; }
        if (!done) {
            c.prog.clear();
            bool any = false;
            for (const ClassTerm& t : cterms[e]) {
                c.prog.insert(c.prog.end(), t.prog.begin(), t.prog.end());
                c.prog.push_back({Q_FOLD, any ? C_YPOW0 + (t.cons - c.last) : C_Y, 0});       // acc = acc * y^(gap) + g_i
                c.last = t.cons;
                any = true;
            }
        }
        c.products = count_products(c.prog);
    }
    for (QPlanClass& c : qp.cls) {
        std::unordered_set<uint32_t> seen;
        for (const Instr& in : c.prog)
            if (in.op == Q_PUSH_COL && seen.insert(in.a).second) {
                c.refs.push_back(in.a);
                if (std::find(qp.refs.begin(), qp.refs.end(), in.a) == qp.refs.end()) qp.refs.push_back(in.a);
            }
    }
    // What the plan executes per proof, in units of one field product over n rows (~10 us at k = 20): a coset transform of a
    // witness-side column ~ 10 of them (the key's own columns are transformed once per key and cached), a remainder its program
    // over H, an inverse transform, and the transforms the two classes that read it already count.
    auto of_key = [](uint32_t ref) { const uint32_t t = ref >> 24; return t == CT_FIXED || t == CT_SIGMA || t == CT_SPECIAL; };
    const double C_T = 10.0;
    qp.cost = 0;
    for (uint32_t e = 0; e <= E; ++e) {
        if (!qp.cls[e].used) continue;
        uint32_t moving = 0;
        for (uint32_t ref : qp.cls[e].refs) moving += !of_key(ref);
        qp.cost += (double)(1u << e) * (moving * C_T + (double)qp.cls[e].products + 2.0);
    }
    for (const QPlanRem& rm : qp.rems) qp.cost += (double)rm.products + C_T + 4.0;
    return ZK_OK;
}
static std::string quotient_plan_key() {
    std::string k;
    for (const char* name : {"ZK_QUOTIENT_SPLIT", "ZK_QUOTIENT_ADDSPLIT", "ZK_QUOTIENT_GROUP", "ZK_QUOTIENT_DAG", "ZK_QUOTIENT_COSTGATE"}) { const char* v = getenv(name); k += v ? v : "-"; k += '|'; }
    return k;
}
static int quotient_plan(const zk_pk* pk, std::shared_ptr<const QuotientPlan>* out, std::string* err) {
    const std::string key = quotient_plan_key();
    if (pk->qplan && pk->qplan->key == key) { *out = pk->qplan; return ZK_OK; }
    const bool split_on = quotient_split_enabled(false, false), add_on = quotient_addsplit_enabled();
    const char* gate_env = getenv("ZK_QUOTIENT_COSTGATE");
    const bool gate = !(gate_env && atoi(gate_env) == 0);
    std::shared_ptr<QuotientPlan> best;
    std::vector<std::pair<bool, bool>> cand;
    if (split_on && add_on) cand.push_back({true, true});
    if (split_on && (gate || !add_on)) cand.push_back({true, false});
    if (!split_on || gate) cand.push_back({false, false});
    for (const auto& c : cand) {
        auto qp = std::make_shared<QuotientPlan>();
        PK_TRY(make_quotient_plan(pk, c.first, c.second, *qp, err));
        if (getenv("ZK_QUOTIENT_TRACE")) {
            fprintf(stderr, "[zk quotient] plan split=%d addsplit=%d dag=%d: cost %.0f, %zu remainders;", (int)qp->split, (int)qp->addsplit, (int)qp->dag, qp->cost, qp->rems.size());
            for (uint32_t e = 0; e <= qp->E; ++e) if (qp->cls[e].used) fprintf(stderr, " class %u: %zu instr, %u products, %zu columns, %u parked (%u live), %u groups;", e, qp->cls[e].prog.size(), qp->cls[e].products, qp->cls[e].refs.size(), qp->cls[e].parked, qp->cls[e].max_live, qp->cls[e].groups);
            fprintf(stderr, "\n");
        }
        if (!best || (gate && qp->cost < best->cost)) best = qp;
        if (!gate) break;
    }
    best->key = key;
    pk->qplan = best;
    *out = best;
    return ZK_OK;
}
// For every advice column: the set of cosets r (bit r) of the extended domain on which some constraint class active there reads it.
static int advice_coset_plan(zk_ctx* ctx, const zk_pk* pk, bool sharded, std::vector<uint32_t>& mask, size_t* key_slots = nullptr) {
    (void)sharded;
    std::shared_ptr<const QuotientPlan> qp;
    {
        std::string err;
        const int rc = quotient_plan(pk, &qp, &err);
        if (rc) return ctx->fail(rc, "%s", err.c_str());
    }
    const uint32_t E = qp->E;
    mask.assign(pk->A, 0u);
    if (E > 5) return ZK_OK;           // more than 32 cosets: no plan (the quotient transforms everything itself)
    std::unordered_map<uint32_t, uint32_t> key_mask;
    for (uint32_t e = 0; e <= E; ++e) {
        uint32_t cosets = 0;
        for (uint32_t r = 0; r < (1u << E); ++r) if ((r & ((1u << (E - e)) - 1u)) == 0) cosets |= 1u << r;
        for (uint32_t ref : qp->cls[e].refs) {
            const uint32_t t = ref >> 24;
            if (t == CT_ADVICE && (ref & 0xFFFFFFu) < pk->A) mask[ref & 0xFFFFFFu] |= cosets;
            else if (t == CT_FIXED || t == CT_SIGMA || t == CT_SPECIAL) key_mask[ref] |= cosets;      // (column of the key, coset) pairs the quotient reads: what the key's coset cache will hold
        }
    }
    if (key_slots) {
        size_t cnt = 0;
        for (const auto& kv : key_mask) cnt += (size_t)__builtin_popcount(kv.second);
        *key_slots = cnt;
    }
    return ZK_OK;
}

// Which advice columns some program over the LAGRANGE domain reads inside a session: the tuples of the lookup arguments, the permutation argument's columns, the
// remainder polynomials of the additive split (quotient plan of the key).  Everything else -- gates only -- is read through coefficient forms (cosets, evaluations).
static int advice_lagrange_readers(zk_ctx* ctx, const zk_pk* pk, std::vector<uint8_t>* need) {
    need->assign(pk->A, 0);
    auto scan = [&](const Prog& g) {
        for (const Instr& in : g)
            if (in.op == Q_PUSH_COL && (in.a >> 24) == CT_ADVICE && (in.a & 0xFFFFFFu) < pk->A) (*need)[in.a & 0xFFFFFFu] = 1;
    };
    for (const auto& lk : pk->lookups) {
        for (const Prog& g : lk.tables) scan(g);
        for (const auto& tuple : lk.inputs) for (const Prog& g : tuple) scan(g);
    }
    for (const auto& pc : pk->perm_cols) if (pc.first == CT_ADVICE && pc.second < pk->A) (*need)[pc.second] = 1;
    std::shared_ptr<const QuotientPlan> qplan;
    std::string err;
    const int rc = quotient_plan(pk, &qplan, &err);
    if (rc) return ctx->fail(rc, "%s", err.c_str());
    for (const QPlanRem& r : qplan->rems) scan(r.prog);
    return ZK_OK;
}

// Which cosets of which advice columns this session computes ahead, during its advice phases (zk_proof::pre_mask, adv_coset).
// Cosets are taken in the order of how many advice columns they serve, as long as the buffers fit what the device can spare:
// free memory (the pool's parked blocks count as free) less everything the session still has to allocate -- Lagrange and
// coefficient forms of the advice columns, of m / phi / Z, one coset buffer per column the quotient reads, h -- less the
// key's own coset cache if it is still to be filled, capped by ZK_ADVICE_COSET_GB (default 64; 0 turns the feature off).
// Sharded sessions do not precompute (their quotient is split by coset over the ranks).
static int plan_advice_cosets(zk_ctx* ctx, zk_proof* pr) {
    pr->pre_planned = true;
    const zk_pk* pk = pr->pk;
    const bool sharded = pr->world > 1 && pr->gather;
    const char* env = getenv("ZK_ADVICE_COSET_GB");
    // A witness that is already on the device leaves no PCIe wait to fill: the phase is bound by the device's arithmetic, and cosets
    // computed ahead only compete with the commitments (measured on the SuperCircuit shape, 60/30/10 witness, resident columns: two
    // cosets ahead 1.54 s per proof, none 1.41 s).  Off for such sessions unless asked for.
    const double cap = (env ? atof(env) : (pr->advice_on_device ? 0.0 : 64.0)) * (double)(1ull << 30);
    if (sharded || cap <= 0 || pk->A == 0) return ZK_OK;
    std::vector<uint32_t> mask;
    size_t key_slots = 0;
    PK_TRY(advice_coset_plan(ctx, pk, sharded, mask, &key_slots));
    const uint32_t E = pk->ext_k - pk->k, R = 1u << E;
    if (E > 5) return ZK_OK;
    const double col_bytes = (double)((size_t)1 << pk->k) * 32.0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return ZK_OK; }
    // columns' worth of buffers still to come, the advice columns' own coset buffers aside (counted below, per choice of cosets):
    // Lagrange + coefficient forms of the advice columns, of m / phi / Z with their temporaries, their coset buffers, h, slack
    // (a witness handed over in place brings its Lagrange forms along)
    const double cols_ahead = (pr->advice_in_place ? 1.0 : 2.0) * pk->A + 3.0 * (2.0 * pk->L + pk->C) + (2.0 * pk->L + pk->C + pk->I + 8.0) + 2.0 * R + 32.0;
    double avail = (double)free_b + (double)ctx->pool_bytes - cols_ahead * col_bytes - 16.0 * (double)(1ull << 30);
    if (pk->part_cache_state < 0 || (pk->part_cache_state == 1 && pk->part_cache_bytes == 0)) {
        // the key's own cosets (fixed, sigma, l_0 ...) are still to be cached by this proof's quotient: reserved, counted from the
        // class programs (a fixed column read only by low-degree gates is cached on two cosets, not on all)
        const double key_need = (double)key_slots * col_bytes;
        const char* kenv = getenv("ZK_PK_COSET_CACHE_GB");
        const double kcap = std::min((kenv ? atof(kenv) : 96.0) * (double)(1ull << 30), (double)ctx->prop.totalGlobalMem / 3.0);
        if (key_need <= kcap) avail -= std::max(0.0, key_need - (double)pk->part_cache_bytes);
    }
    const double budget = std::min(cap, avail);
    std::vector<std::pair<uint32_t, uint32_t>> by_count;          // (columns served, coset)
    for (uint32_t r = 0; r < R; ++r) {
        uint32_t cnt = 0;
        for (uint32_t m_ : mask) cnt += m_ >> r & 1u;
        if (cnt) by_count.push_back({cnt, r});
    }
    std::sort(by_count.begin(), by_count.end(), [](const auto& a, const auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
    uint32_t chosen = 0;
    double used = 0;
    for (const auto& cr : by_count) {
        uint32_t rest = 0;                       // advice columns the quotient still has to transform itself (one coset buffer each)
        for (uint32_t m_ : mask) rest += (m_ & ~(chosen | 1u << cr.second)) != 0;
        if (used + cr.first * col_bytes > cap || used + (cr.first + rest) * col_bytes > avail) break;
        used += cr.first * col_bytes;
        chosen |= 1u << cr.second;
    }
    if (getenv("ZK_PROVER_TRACE")) {
        fprintf(stderr, "[zk prover] advice coset plan: free %.1f + pooled %.1f GiB, reserve %.1f GiB, budget %.1f GiB, cosets by columns served:", free_b / 1073741824.0, ctx->pool_bytes / 1073741824.0,
                (cols_ahead * col_bytes) / 1073741824.0 + 16.0, budget / 1073741824.0);
        for (const auto& cr : by_count) fprintf(stderr, " r%u:%u", cr.second, cr.first);
        fprintf(stderr, " -> chosen 0x%x\n", chosen);
    }
    if (!chosen) return ZK_OK;
    pr->pre_mask.assign(pk->A, 0u);
    for (uint32_t c = 0; c < pk->A; ++c) pr->pre_mask[c] = mask[c] & chosen;
    pr->adv_coset.resize(R);
    for (auto& v : pr->adv_coset) v.resize(pk->A);
    if (getenv("ZK_PROVER_TRACE")) fprintf(stderr, "[zk prover] advice cosets computed ahead: cosets 0x%x, %.1f GiB\n", chosen, used / (double)(1ull << 30));
    return ZK_OK;
}

// Host only (no device), for tests: the degree-class assembly of constraint programs with shared intermediates, exactly as
// zk_proof_finish performs it (TmpSplit).  words: 3 per instruction, the `count` programs back to back (lens[i] instructions
// each); cls[i] < classes.  Output: the class programs back to back (out_lens[e] instructions each), every constraint followed
// by the marker {Q_FOLD, i, 0} so that a test can tell the constraints' values apart.  *conflict = 1 when the programs reuse a
// slot across constraints (the prover then keeps a single class).
int zk_host_split_programs(const uint32_t* words, const uint32_t* lens, const uint32_t* cls, uint32_t count, uint32_t classes,
                           uint32_t* out_words, size_t out_cap_words, uint32_t* out_lens, int* conflict) {
    if (!words || !lens || !cls || !out_lens || !conflict || classes == 0) return ZK_ERR_INVALID_ARG;
    std::vector<Prog> cons(count);
    size_t at = 0;
    for (uint32_t i = 0; i < count; ++i) {
        if (cls[i] >= classes) return ZK_ERR_INVALID_ARG;
        for (uint32_t j = 0; j < lens[i]; ++j, ++at) cons[i].push_back({words[3 * at], words[3 * at + 1], words[3 * at + 2]});
    }
    *conflict = tmp_slots_conflict(cons) ? 1 : 0;
    std::vector<Prog> progs(classes);
    TmpSplit tmps(classes);
    for (uint32_t i = 0; i < count; ++i) {
        tmps.append(cons[i], cls[i], progs[cls[i]]);
        progs[cls[i]].push_back({Q_FOLD, i, 0});
    }
    if (tmps.conflict) *conflict = 1;
    size_t total = 0;
    for (uint32_t e = 0; e < classes; ++e) { out_lens[e] = (uint32_t)progs[e].size(); total += progs[e].size(); }
    if (!out_words) return ZK_OK;
    if (3 * total > out_cap_words) return ZK_ERR_INVALID_ARG;
    size_t w = 0;
    for (const Prog& pg : progs) for (const Instr& in : pg) { out_words[w++] = in.op; out_words[w++] = in.a; out_words[w++] = in.b; }
    return ZK_OK;
}

// Host only (no device), for tests: the additive split of ONE constraint program over the degree classes 0 .. E exactly as
// zk_proof_finish applies it (additive_split above).  Output: the pieces back to back (out_lens[j] instructions of class
// out_cls[j] each), *num_pieces of them -- one piece, the program itself in the class of its degree, when it is not split.
int zk_host_additive_split(const uint32_t* words, uint32_t num_instr, uint32_t E, uint32_t* out_words, size_t out_cap_words, uint32_t* out_cls, uint32_t* out_lens,
                           uint32_t cap_pieces, uint32_t* num_pieces) {
    if (!words || !num_pieces || E > 8) return ZK_ERR_INVALID_ARG;
    Prog g(num_instr);
    for (uint32_t j = 0; j < num_instr; ++j) g[j] = {words[3 * j], words[3 * j + 1], words[3 * j + 2]};
    std::vector<int> tmp_deg;
    const int dg = program_degree(g, &tmp_deg);
    if (dg < 0) return ZK_ERR_INVALID_ARG;
    std::vector<std::pair<uint32_t, Prog>> parts;
    const uint32_t whole = class_of_degree(dg, E);
    if (!(whole > 0 && additive_split(g, E, whole, parts))) { parts.clear(); parts.push_back({whole, g}); }
    *num_pieces = (uint32_t)parts.size();
    if (!out_words) return ZK_OK;
    if (!out_cls || !out_lens || parts.size() > cap_pieces) return ZK_ERR_INVALID_ARG;
    size_t w = 0;
    for (size_t j = 0; j < parts.size(); ++j) {
        out_cls[j] = parts[j].first;
        out_lens[j] = (uint32_t)parts[j].second.size();
        if (w + 3 * parts[j].second.size() > out_cap_words) return ZK_ERR_INVALID_ARG;
        for (const Instr& in : parts[j].second) { out_words[w++] = in.op; out_words[w++] = in.a; out_words[w++] = in.b; }
    }
    return ZK_OK;
}

int zk_host_group_terms(const uint32_t* words, const uint32_t* lens, const uint32_t* cons, uint32_t count, uint32_t K, uint32_t* out_words, size_t out_cap_words, uint32_t* out_instr) {
    if (!words || !lens || !cons || !out_instr) return ZK_ERR_INVALID_ARG;
    std::vector<ClassTerm> terms(count);
    size_t at = 0;
    for (uint32_t t = 0; t < count; ++t) {
        if (cons[t] >= K) return ZK_ERR_INVALID_ARG;
        terms[t].cons = cons[t];
        terms[t].prog.resize(lens[t]);
        for (uint32_t j = 0; j < lens[t]; ++j, ++at) terms[t].prog[j] = {words[3 * at], words[3 * at + 1], words[3 * at + 2]};
    }
    Prog out;
    if (!assemble_grouped(terms, K, out)) return ZK_ERR_UNSUPPORTED;
    *out_instr = (uint32_t)out.size();
    if (!out_words) return ZK_OK;
    if (out_cap_words < 3 * out.size()) return ZK_ERR_INVALID_ARG;
    for (size_t j = 0; j < out.size(); ++j) { out_words[3 * j] = out[j].op; out_words[3 * j + 1] = out[j].a; out_words[3 * j + 2] = out[j].b; }
    return ZK_OK;
}

// Host only, for tests: compile_class (class_compile.hpp) over the terms of one class.  out_stats[0..5] = graph nodes, values parked,
// parking slots alive at once, products, factor groups, stack depth; *out_last = the constraint index the program's sum is aligned to
// (the caller scales by y^(K-1-last)).
int zk_host_compile_class(const uint32_t* words, const uint32_t* lens, const uint32_t* cons, uint32_t count, uint32_t K, uint32_t* out_words, size_t out_cap_words, uint32_t* out_instr,
                          uint32_t* out_last, uint32_t* out_stats) {
    if (!words || !lens || !cons || !out_instr || !out_last) return ZK_ERR_INVALID_ARG;
    std::vector<ClassTerm> terms(count);
    size_t at = 0;
    for (uint32_t t = 0; t < count; ++t) {
        if (cons[t] >= K) return ZK_ERR_INVALID_ARG;
        terms[t].cons = cons[t];
        terms[t].prog.resize(lens[t]);
        for (uint32_t j = 0; j < lens[t]; ++j, ++at) terms[t].prog[j] = {words[3 * at], words[3 * at + 1], words[3 * at + 2]};
    }
    Prog out;
    ClassCompileStats st;
    if (!compile_class(terms, K, out, out_last, &st)) return ZK_ERR_UNSUPPORTED;
    *out_instr = (uint32_t)out.size();
    if (out_stats) { out_stats[0] = st.nodes; out_stats[1] = st.parked; out_stats[2] = st.max_live; out_stats[3] = st.products; out_stats[4] = st.groups; out_stats[5] = (uint32_t)st.depth; }
    if (!out_words) return ZK_OK;
    if (out_cap_words < 3 * out.size()) return ZK_ERR_INVALID_ARG;
    for (size_t j = 0; j < out.size(); ++j) { out_words[3 * j] = out[j].op; out_words[3 * j + 1] = out[j].a; out_words[3 * j + 2] = out[j].b; }
    return ZK_OK;
}

// Host only (no device, no SRS): the quotient plan zk_proof_finish will follow for a constraint system -- `cs_blob` is the
// constraint-system part of a key blob (the column data may be missing).  out_summary: [0] E = ext_k - k, [1] K constraints,
// [2] degree classes on, [3] additive split on, [4] compiled through the expression graph, [5] remainder polynomials, [6] the cost
// estimate the candidates were compared by, [7] columns read; then 8 words per class e <= E: used, instructions, products, columns
// read, values parked, parking slots alive at once, factor groups, last.  class_index <= E with out_words: that class's program.
static int write_plan_summary(const std::shared_ptr<const QuotientPlan>& qp, uint32_t* out_summary, size_t cap_summary, uint32_t class_index, uint32_t* out_words, size_t out_cap_words, uint32_t* out_instr);
int zk_host_quotient_plan(const void* cs_blob, size_t blob_len, uint32_t* out_summary, size_t cap_summary, uint32_t class_index, uint32_t* out_words, size_t out_cap_words, uint32_t* out_instr) {
    if (!cs_blob || !out_summary) return ZK_ERR_INVALID_ARG;
    Reader r{(const uint8_t*)cs_blob, blob_len};
    zk_pk pk;
    std::string err;
    int rc = parse_cs(r, &pk, blob_len, false, &err);
    if (rc) { fprintf(stderr, "zk_host_quotient_plan: %s\n", err.c_str()); return rc; }
    std::shared_ptr<const QuotientPlan> qp;
    rc = quotient_plan(&pk, &qp, &err);
    if (rc) { fprintf(stderr, "zk_host_quotient_plan: %s\n", err.c_str()); return rc; }
    return write_plan_summary(qp, out_summary, cap_summary, class_index, out_words, out_cap_words, out_instr);
}
// The plan of a key (what zk_proof_finish follows for it under the knobs in force): same summary as zk_host_quotient_plan.
int zk_pk_quotient_plan(zk_ctx* ctx, const zk_pk* pk, uint32_t* out_summary, size_t cap_summary) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pk && out_summary, "null pointer");
    std::shared_ptr<const QuotientPlan> qp;
    std::string err;
    const int rc = quotient_plan(pk, &qp, &err);
    if (rc) return ctx->fail(rc, "%s", err.c_str());
    if (write_plan_summary(qp, out_summary, cap_summary, 0xFFFFFFFFu, nullptr, 0, nullptr)) return ctx->fail(ZK_ERR_INVALID_ARG, "zk_pk_quotient_plan: the summary needs %u words", 8 + 8 * (qp->E + 1));
    return ZK_OK;
}
static int write_plan_summary(const std::shared_ptr<const QuotientPlan>& qp, uint32_t* out_summary, size_t cap_summary, uint32_t class_index, uint32_t* out_words, size_t out_cap_words, uint32_t* out_instr) {
    if (cap_summary < 8 + 8 * (size_t)(qp->E + 1)) return ZK_ERR_INVALID_ARG;
    out_summary[0] = qp->E; out_summary[1] = qp->K; out_summary[2] = qp->split; out_summary[3] = qp->addsplit; out_summary[4] = qp->dag;
    out_summary[5] = (uint32_t)qp->rems.size(); out_summary[6] = (uint32_t)std::min(qp->cost, 4.0e9); out_summary[7] = (uint32_t)qp->refs.size();
    for (uint32_t e = 0; e <= qp->E; ++e) {
        const QPlanClass& c = qp->cls[e];
        uint32_t* o = out_summary + 8 + 8 * e;
        o[0] = c.used; o[1] = (uint32_t)c.prog.size(); o[2] = c.products; o[3] = (uint32_t)c.refs.size(); o[4] = c.parked; o[5] = c.max_live; o[6] = c.groups; o[7] = c.last;
    }
    if (out_instr && class_index <= qp->E) {
        const Prog& g = qp->cls[class_index].prog;
        *out_instr = (uint32_t)g.size();
        if (out_words) {
            if (out_cap_words < 3 * g.size()) return ZK_ERR_INVALID_ARG;
            for (size_t j = 0; j < g.size(); ++j) { out_words[3 * j] = g[j].op; out_words[3 * j + 1] = g[j].a; out_words[3 * j + 2] = g[j].b; }
        }
    }
    return ZK_OK;
}

// Everything after the advice phases: lookups, permutation, quotient, evaluations, multi-open.
// Consumes the session (it is freed whether or not the call succeeds).
int zk_proof_finish(zk_ctx* ctx, zk_proof* pr_raw, void* h_proof, size_t proof_cap, size_t* proof_len) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    PoolScope pool_scope(ctx);
    std::unique_ptr<zk_proof> pr(pr_raw);
    ZK_REQUIRE(ctx, pr && h_proof && proof_len, "null pointer");
    const zk_pk* pk = pr->pk;
    if (pr->phase != pk->num_phases) return ctx->fail(ZK_ERR_INVALID_ARG, "only %u of %u advice phases were committed", pr->phase, pk->num_phases);
    const zk_srs* srs = pk->srs;
    const uint32_t k = pk->k, ext_k = pk->ext_k;
    const size_t n = (size_t)1 << k, ne = (size_t)1 << ext_k;
    host::XorShiftRng& rng = pr->rng;
    host::Transcript& tr = pr->tr;
    std::vector<DevBuf>&inst_lag = pr->inst_lag, &inst_coeff = pr->inst_coeff, &adv_lag = pr->adv_lag;
    std::vector<DevBuf>& adv_coeff = pr->adv_coeff;     // computed during the advice phases, in the shadow of the uploads
    const F4 one = host::fr_one();
    StageTrace trace(ctx);
    Env lag{pk, nullptr, &adv_lag, &inst_lag, nullptr, nullptr, nullptr, one, one, one, one, {}, pr->challenges};
    lag.theta = tr.squeeze();
    // ---- more cosets of the advice columns, ahead of the quotient: what the advice phases could not fit (or afford) is
    // computed NOW on the auxiliary stream, beside the lookup / permutation / grand-sum stages below -- hash joins, scans,
    // batch inversions and chains of small-valued commitments that leave most of the device idle.  Same plan, same budget
    // rule as in the advice phase, with the memory that is free at this point; the quotient waits for the stream before its
    // first coset.  OFF by default (ZK_ADVICE_COSET_LATE_GB=<GiB> turns it on): measured on the SuperCircuit shape with 48 GiB
    // (tools/gpu_r3v.sh) the quotient's transform stage drops from 237 to 149 ms and the lookup stage beside which the transforms
    // run grows from 46 to 118 ms -- 1.437 -> 1.424 s for 25 GiB of device memory: these stages are not idle enough to hide them.
    bool late_cosets = false;
    // whatever way this function is left, the auxiliary stream must be through with the session's buffers before they go back to the pool
    struct AuxJoin { zk_ctx* c; bool* on; ~AuxJoin() { if (*on && c->stream_aux) (void)hipStreamSynchronize(c->stream_aux); } } aux_join{ctx, &late_cosets};
    {
        const bool sharded_ = pr->world > 1 && pr->gather;
        const char* env = getenv("ZK_ADVICE_COSET_LATE_GB");
        const double cap = (env ? atof(env) : 0.0) * (double)(1ull << 30);
        const uint32_t E_ = ext_k - k, R_ = 1u << E_;
        if (!sharded_ && cap > 0 && pk->A && E_ <= 5 && ctx->ensure_aux()) {
            std::vector<uint32_t> mask;
            size_t key_slots = 0;
            PK_TRY(advice_coset_plan(ctx, pk, false, mask, &key_slots));
            if (pr->adv_coset.empty()) { pr->adv_coset.resize(R_); for (auto& v : pr->adv_coset) v.resize(pk->A); }
            const double col_bytes = (double)n * 32.0;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
            // still to come: m / phi / Z in Lagrange and coefficient form with their temporaries, their coset buffers, h, slack;
            // the key's own cosets if its cache is still empty
            double avail = (double)free_b + (double)ctx->pool_bytes - (4.0 * (2.0 * pk->L + pk->C) + (2.0 * pk->L + pk->C + pk->I + 8.0) + 2.0 * R_ + 32.0) * col_bytes - 24.0 * (double)(1ull << 30);
            if (pk->part_cache_state < 0 || (pk->part_cache_state == 1 && pk->part_cache_bytes == 0)) avail -= (double)key_slots * col_bytes;
            std::vector<std::pair<uint32_t, uint32_t>> by_count;          // (columns still to transform, coset)
            for (uint32_t r = 0; r < R_; ++r) {
                uint32_t cnt = 0;
                for (uint32_t c = 0; c < pk->A; ++c) cnt += (mask[c] >> r & 1u) && !pr->adv_coset[r][c].p && adv_coeff[c].p;
                if (cnt) by_count.push_back({cnt, r});
            }
            std::sort(by_count.begin(), by_count.end(), [](const auto& a, const auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
            double used = 0;
            hipStream_t main_stream = ctx->stream;
            struct Back { zk_ctx* c; hipStream_t s; ~Back() { c->stream = s; } } back{ctx, main_stream};
            const Fr w_ext = fr_root_of_unity(ext_k);
            for (const auto& cr : by_count) {
                if (used + cr.first * col_bytes > std::min(cap, avail)) continue;          // a smaller coset further down may still fit
                Fr g = fr_zeta();
                for (uint32_t i = 0; i < cr.second; ++i) g = g * w_ext;
                std::vector<const void*> csrc;
                std::vector<void*> cdst;
                bool ok = true;
                for (uint32_t c = 0; c < pk->A && ok; ++c) {
                    if (!(mask[c] >> cr.second & 1u) || pr->adv_coset[cr.second][c].p || !adv_coeff[c].p) continue;
                    DevBuf& slot = pr->adv_coset[cr.second][c];
                    if (!slot.alloc(n * 32)) { ok = false; break; }
                    csrc.push_back(adv_coeff[c].p);
                    cdst.push_back(slot.p);
                }
                if (!late_cosets) {          // first batch: the stream starts behind everything the main stream has enqueued (pooled blocks)
                    ZK_HIP(ctx, hipEventRecord(ctx->ev_aux, main_stream));
                    ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream_aux, ctx->ev_aux, 0));
                }
                ctx->stream = ctx->stream_aux;
                late_cosets = true;
                if (!csrc.empty()) PK_TRY(zk_coeff_to_coset_batch(ctx, csrc.data(), k, &g, cdst.data(), csrc.size()));
                ctx->stream = main_stream;
                used += csrc.size() * col_bytes;
                if (!ok) break;
            }
            if (late_cosets) {
                ZK_HIP(ctx, hipEventRecord(ctx->ev_aux, ctx->stream_aux));
                if (getenv("ZK_PROVER_TRACE")) fprintf(stderr, "[zk prover] advice cosets computed beside the lookup / permutation stages: %.1f GiB\n", used / (double)(1ull << 30));
            }
        }
    }

    // ---- lookups, round 1 (mv_lookup::prover::Argument::prepare): theta-compressed table and input tuples,
    // multiplicities m over ALL input tuples of the argument.  Every lookup is enqueued back to back
    // (compression programs, device hash join); one download of the status words, one pipelined batch
    // of commits.  The unusable rows of m stay zero, as upstream leaves them.
    std::vector<std::vector<DevBuf>> lk_f(pk->L);
    std::vector<DevBuf> lk_t(pk->L), lk_m(pk->L), lk_phi(pk->L);
    std::vector<uint8_t> same_table(pk->L, 0);            // lookup l reads the table of the lookup this rank worked on before it (lk_t lives at table_owner[l])
    std::vector<uint32_t> table_owner(pk->L, 0);
    // Sharded sessions (round 6): the lookup arguments are split over the ranks like their commitments -- rank r compresses the tuples, counts the multiplicities and forms the
    // running sum of arguments r, r + world, ... only; the Lagrange forms of m and phi (what the additive split's remainders, the coefficient forms and with them every later
    // stage read) are all-gathered device to device behind each of the two rounds.  The blinding rows are drawn for every argument on every rank (one RNG sequence).
    // ZK_SHARD_LOOKUPS=0: every rank works on every argument, as before.
    const bool shard_args = pr->world > 1 && pr->gather && pk->L >= pr->world && !(getenv("ZK_SHARD_LOOKUPS") && atoi(getenv("ZK_SHARD_LOOKUPS")) == 0);
    std::vector<uint32_t> act;                             // the arguments this rank works on, in order
    for (uint32_t l = 0; l < pk->L; ++l) if (!shard_args || l % pr->world == pr->rank) act.push_back(l);
    // all-gather of columns owned round-robin (column l by rank l % world): a rank's columns packed into one send buffer, several groups of `world` per exchange
    auto exchange_owned = [&](std::vector<DevBuf>& cols) -> int {
        const size_t total = cols.size(), W = pr->world, groups_total = (total + W - 1) / W;
        static const size_t xg_knob = getenv("ZK_SHARD_EXCHANGE_GROUPS") ? (size_t)atol(getenv("ZK_SHARD_EXCHANGE_GROUPS")) : 16;
        const size_t XG = std::max<size_t>(1, std::min(xg_knob, groups_total));
        DevBuf gbuf, sbuf;
        if (!gbuf.alloc(W * XG * n * 32) || !sbuf.alloc(XG * n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
        std::vector<uint8_t> hs, hr;
        for (size_t grp0 = 0; grp0 < groups_total; grp0 += XG) {
            const size_t gcnt = std::min(XG, groups_total - grp0);
            for (size_t g = 0; g < gcnt; ++g) {
                const size_t mine_c = (grp0 + g) * W + pr->rank;
                if (mine_c < total) ZK_HIP(ctx, hipMemcpyAsync((char*)sbuf.p + g * n * 32, cols[mine_c].p, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
                else ZK_HIP(ctx, hipMemsetAsync((char*)sbuf.p + g * n * 32, 0, n * 32, ctx->stream));
            }
            if (pr->gather_dev) {
                PK_TRY(zk_ctx_sync(ctx));
                if (pr->gather_dev(pr->gather_dev_user, sbuf.p, gcnt * n * 32, gbuf.p)) return ctx->fail(ZK_ERR_INVALID_ARG, "sharded session: device all-gather callback failed");
            } else {
                hs.resize(gcnt * n * 32); hr.resize(W * gcnt * n * 32);
                PK_TRY(zk_d2h(ctx, hs.data(), sbuf.p, hs.size()));
                if (pr->gather(pr->gather_user, hs.data(), hs.size(), hr.data())) return ctx->fail(ZK_ERR_INVALID_ARG, "sharded session: all-gather callback failed");
                PK_TRY(zk_h2d(ctx, gbuf.p, hr.data(), hr.size()));
            }
            for (size_t g = 0; g < gcnt; ++g)
                for (uint32_t q_ = 0; q_ < W; ++q_) {
                    const size_t c_ = (grp0 + g) * W + q_;
                    if (q_ == pr->rank || c_ >= total) continue;
                    ZK_HIP(ctx, hipMemcpyAsync(cols[c_].p, (char*)gbuf.p + ((size_t)q_ * gcnt + g) * n * 32, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
                }
            PK_TRY(zk_ctx_sync(ctx));                      // gbuf / sbuf are reused by the next chunk (and freed at the end)
        }
        return ZK_OK;
    };
    if (pk->L) {
        Prog prev_table;
        DevBuf status;
        if (!status.alloc((size_t)pk->L * 4)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
        ZK_HIP(ctx, hipMemsetAsync(status.p, 0xFF, (size_t)pk->L * 4, ctx->stream));
        std::vector<const void*> mptrs(pk->L);
        for (uint32_t l = 0; l < pk->L; ++l) {            // every m exists on every rank (the ones of other ranks arrive below)
            if (!lk_m[l].alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
            mptrs[l] = lk_m[l].p;
        }
        uint32_t prev_l = 0;
        bool have_prev = false;
        for (const uint32_t l : act) {
            const auto& lk = pk->lookups[l];
            PB pt;
            push_compressed(pt, lk.tables); pt.fold(C_ONE);
            // consecutive arguments into the same table (chunk_lookups() splits a table's inputs over as many arguments as the
            // degree bound needs; the EVM circuit's 80-odd lookups go into a dozen tables): one compressed table, one hash
            static const bool share_tables = !(getenv("ZK_LOOKUP_SHARE") && atoi(getenv("ZK_LOOKUP_SHARE")) == 0);      // measurement knob
            same_table[l] = share_tables && have_prev && pt.g.size() == prev_table.size() && memcmp(pt.g.data(), prev_table.data(), pt.g.size() * sizeof(Instr)) == 0;
            table_owner[l] = same_table[l] ? table_owner[prev_l] : l;
            prev_table = pt.g;
            prev_l = l; have_prev = true;
            if (!same_table[l]) {
                if (!lk_t[l].alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
                PK_TRY(run_program(ctx, lag, pt.g, lk_t[l].p));
            }
            lk_f[l].resize(lk.inputs.size());
            std::vector<const Fr*> fptrs;
            for (size_t a = 0; a < lk.inputs.size(); ++a) {
                PB pf;
                push_compressed(pf, lk.inputs[a]); pf.fold(C_ONE);
                if (!lk_f[l][a].alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
                PK_TRY(run_program(ctx, lag, pf.g, lk_f[l][a].p));
                fptrs.push_back(lk_f[l][a].fr());
            }
            PK_TRY(lookup_multiplicities_enqueue(ctx, fptrs.data(), fptrs.size(), lk_t[table_owner[l]].fr(), pk->u, lk_m[l].fr(), n, (uint32_t*)status.p + l, same_table[l]));
        }
        std::vector<uint32_t> st(pk->L);
        PK_TRY(zk_d2h(ctx, st.data(), status.p, (size_t)pk->L * 4));
        for (uint32_t l = 0; l < pk->L; ++l)
            if (st[l] != 0xFFFFFFFFu) return ctx->fail(ZK_ERR_INVALID_ARG, "lookup %u: input at row %u is not in the table (witness does not satisfy the circuit)", l, st[l]);
        trace.mark("  lookup: m (all lookups)");
        std::vector<G1Affine> coms(pk->L);
        PK_TRY(sharded_commit(ctx, pr.get(), srs, 1, mptrs.data(), pk->L, n, coms.data(), 1));      // multiplicities are small counts
        for (const G1Affine& com : coms) tr.write_point(com);
        if (shard_args) PK_TRY(exchange_owned(lk_m));
    }
    trace.mark("lookup m");
    lag.lk_m = &lk_m;
    lag.beta = tr.squeeze();
    lag.gamma = tr.squeeze();
    {   // beta * delta^j for the permutation numerators;  delta = 7^(2^28)
        F4 delta = host::fr_pow(host::fr_from_u64(7), 1ull << 28), cur = lag.beta;
        for (uint32_t j = 0; j < pk->P; ++j) { lag.beta_delta.push_back(cur); cur = host::fr_mul(cur, delta); }
    }

    // ---- permutation grand products
    std::vector<DevBuf> pz_lag(pk->C);
    {
        // All chunks are enqueued back to back (ratios -> unscaled running products); the chunks are
        // chained afterwards with one small download: Z_c = Z_c^raw * prod_{c' < c} Z_c'^raw(omega^u).
        DevBuf num, den, tails_d;
        if (!num.alloc(n * 32) || !den.alloc(n * 32) || !tails_d.alloc((size_t)pk->C * 32 + 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
        for (uint32_t c = 0; c < pk->C; ++c) {
            PB pn, pd;
            const uint32_t j0 = c * pk->chunk, j1 = std::min(pk->P, j0 + pk->chunk);
            for (uint32_t j = j0; j < j1; ++j) {
                push_perm_col(pn, pk->perm_cols[j]); pn.col(CT_SPECIAL, SP_X).mulc(C_DELTA0 + j).op(Q_ADD).addc(C_GAMMA);
                if (j > j0) pn.op(Q_MUL);
                push_perm_col(pd, pk->perm_cols[j]); pd.col(CT_SIGMA, j).mulc(C_BETA).op(Q_ADD).addc(C_GAMMA);
                if (j > j0) pd.op(Q_MUL);
            }
            pn.fold(C_ONE); pd.fold(C_ONE);
            PK_TRY(run_program(ctx, lag, pn.g, num.p));
            PK_TRY(run_program(ctx, lag, pd.g, den.p));
            PK_TRY(zk_fr_batch_invert(ctx, den.p, n));
            PK_TRY(zk_field_vec_op(ctx, ZK_FIELD_FR, ZK_OP_MUL, num.p, den.p, num.p, n));
            if (!pz_lag[c].alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
            PK_TRY(zk_fr_prefix_product(ctx, num.p, pz_lag[c].p, n));      // z[0] = 1, z[i+1] = z[i] * ratio[i]
            ZK_HIP(ctx, hipMemcpyAsync((char*)tails_d.p + (size_t)c * 32, (char*)pz_lag[c].p + (size_t)pk->u * 32, 32, hipMemcpyDeviceToDevice, ctx->stream));
        }
        std::vector<F4> tails(pk->C);
        if (pk->C) PK_TRY(zk_d2h(ctx, tails.data(), tails_d.p, (size_t)pk->C * 32));
        trace.mark("  perm: running products");
        std::vector<F4> blind((size_t)pk->C * pk->bf);
        std::vector<const void*> zptrs(pk->C);
        F4 start = one;
        for (uint32_t c = 0; c < pk->C; ++c) {
            if (c) PK_TRY(zk_fr_scale(ctx, pz_lag[c].p, &start, n));       // Z_c(1) = Z_{c-1}(omega^u)
            start = host::fr_mul(start, tails[c]);
            for (uint32_t r_ = 0; r_ < pk->bf; ++r_) blind[(size_t)c * pk->bf + r_] = rng.next_fr();
            ZK_HIP(ctx, hipMemcpyAsync((char*)pz_lag[c].p + (n - pk->bf) * 32, blind.data() + (size_t)c * pk->bf, (size_t)pk->bf * 32, hipMemcpyHostToDevice, ctx->stream));
            zptrs[c] = pz_lag[c].p;
        }
        trace.mark("  perm: chain + blind");
        std::vector<G1Affine> coms(pk->C);
        PK_TRY(sharded_commit(ctx, pr.get(), srs, 1, zptrs.data(), pk->C, n, coms.data(), 2));      // running products stay constant over every stretch of rows without copies
        trace.mark("  perm: commits");
        for (const G1Affine& com : coms) tr.write_point(com);
        if (pk->C && !host::fr_eq(start, one)) return ctx->fail(ZK_ERR_INVALID_ARG, "permutation argument does not close: copy constraints are not satisfied by the witness");
    }
    trace.mark("permutation Z");
    // ---- lookups, round 2 (Prepared::commit_grand_sum): phi[0] = 0, phi[i+1] = phi[i] + sum_a 1/(f_a[i]+beta) - m[i]/(t[i]+beta),
    // the last bf rows random; enqueued back to back with one closing check and one commit batch
    if (pk->L) {
        // ONE batch inversion for all arguments of the proof: every (t + beta) and (f_a + beta) column is written into one buffer --
        // slot list below; an argument that reads the table of the one before it has no table slot of its own --, inverted by a
        // single call (a batch inversion is one field inversion per wave behind two passes of products: a hundred calls of
        // 2-3 M elements each were a hundred inversion latencies, 16 ms on the SuperCircuit shape), then every argument forms
        // g = sum_a 1 / (f_a + beta) - m / (t + beta) and its prefix sum.  When the buffer would exceed a quarter of the free device
        // memory the arguments are processed in several such batches.
        std::vector<size_t> slot0(pk->L), tslot(pk->L), nslots(pk->L);
        DevBuf g, closing_d;
        if (!g.alloc(n * 32) || !closing_d.alloc((size_t)pk->L * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
        std::vector<F4> blind((size_t)pk->L * pk->bf);
        for (size_t q = 0; q < blind.size(); ++q) blind[q] = rng.next_fr();            // argument by argument, row by row: the order they were always drawn in, on every rank
        std::vector<const void*> pptrs(pk->L);
        for (uint32_t l = 0; l < pk->L; ++l) {            // every phi exists on every rank (the ones of other ranks arrive below)
            if (!lk_phi[l].alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
            pptrs[l] = lk_phi[l].p;
        }
        ZK_HIP(ctx, hipMemsetAsync(closing_d.p, 0, (size_t)pk->L * 32, ctx->stream));
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        const size_t max_slots = std::max<size_t>(4, (free_b + ctx->pool_bytes) / 4 / (n * 32));
        const size_t NA = act.size();
        for (size_t j0 = 0; j0 < NA;) {
            // arguments act[j0 .. j1) share one inversion; an argument is never separated from the owner of its table
            size_t slots = 0;
            size_t j1 = j0;
            while (j1 < NA) {
                const uint32_t l1 = act[j1];
                const size_t need = lk_f[l1].size() + (same_table[l1] && j1 > j0 ? 0 : 1);
                if (j1 > j0 && slots + need > max_slots && !same_table[l1]) break;
                slot0[l1] = slots;
                tslot[l1] = (same_table[l1] && j1 > j0) ? tslot[act[j1 - 1]] : slots;
                nslots[l1] = need;
                slots += need;
                ++j1;
            }
            DevBuf inv;
            if (!inv.alloc(slots * n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
            {   // t + beta and every f_a + beta of the batch, into their slots: ONE kind of launch for all of them (fr_add_const_many; each used to be a program of its own)
                std::vector<const void*> asrc;
                std::vector<void*> adst;
                for (size_t j = j0; j < j1; ++j) {
                    const uint32_t l = act[j];
                    const size_t N = lk_f[l].size();
                    const bool own_t = tslot[l] == slot0[l];
                    for (size_t a = own_t ? 0 : 1; a <= N; ++a) {
                        const size_t slot = a == 0 ? tslot[l] : slot0[l] + (own_t ? a : a - 1);
                        asrc.push_back(a == 0 ? lk_t[table_owner[l]].p : lk_f[l][a - 1].p);
                        adst.push_back((char*)inv.p + slot * n * 32);
                    }
                }
                PK_TRY(fr_add_const_many(ctx, asrc.data(), adst.data(), asrc.size(), &lag.beta, n));
            }
            PK_TRY(zk_fr_batch_invert(ctx, inv.p, slots * n));
            for (size_t j = j0; j < j1; ++j) {
                const uint32_t l = act[j];
                DevBuf& phi = lk_phi[l];
                const size_t N = lk_f[l].size();
                const bool own_t = tslot[l] == slot0[l];
                const char* inv_t = (const char*)inv.p + tslot[l] * n * 32;
                const char* inv_f = (const char*)inv.p + (slot0[l] + (own_t ? 1 : 0)) * n * 32;
                // g = sum_a inv_f_a - m * inv_t
                PK_TRY(zk_field_vec_op(ctx, ZK_FIELD_FR, ZK_OP_MUL, lk_m[l].p, inv_t, g.p, n));
                PK_TRY(zk_field_vec_op(ctx, ZK_FIELD_FR, ZK_OP_SUB, inv_f, g.p, g.p, n));
                for (size_t a = 1; a < N; ++a) PK_TRY(zk_field_vec_op(ctx, ZK_FIELD_FR, ZK_OP_ADD, g.p, inv_f + a * n * 32, g.p, n));
                PK_TRY(zk_fr_prefix_sum(ctx, g.p, phi.p, n));                       // phi[0] = 0, phi[i+1] = phi[i] + g[i]
                ZK_HIP(ctx, hipMemcpyAsync((char*)closing_d.p + (size_t)l * 32, (char*)phi.p + (size_t)pk->u * 32, 32, hipMemcpyDeviceToDevice, ctx->stream));
                ZK_HIP(ctx, hipMemcpyAsync((char*)phi.p + (n - pk->bf) * 32, blind.data() + (size_t)l * pk->bf, (size_t)pk->bf * 32, hipMemcpyHostToDevice, ctx->stream));
                lk_f[l].clear();                                                    // f, t are not needed again (the quotient recomputes them on its cosets)
                if (j + 1 == NA || !same_table[act[j + 1]]) lk_t[table_owner[l]].release();      // the table's last reader
            }
            j0 = j1;
        }
        std::vector<F4> closing(pk->L);
        PK_TRY(zk_d2h(ctx, closing.data(), closing_d.p, (size_t)pk->L * 32));
        for (uint32_t l = 0; l < pk->L; ++l)
            if (!host::fr_is_zero(closing[l])) return ctx->fail(ZK_ERR_INVALID_ARG, "lookup %u: grand sum does not close", l);
        trace.mark("  lookup: phi (all lookups)");
        std::vector<G1Affine> coms(pk->L);
        PK_TRY(sharded_commit(ctx, pr.get(), srs, 1, pptrs.data(), pk->L, n, coms.data(), 3));      // running sums: mostly equal increments (runs.hip)
        for (const G1Affine& com : coms) tr.write_point(com);
        if (shard_args) PK_TRY(exchange_owned(lk_phi));
    }
    trace.mark("lookup phi");
    // ---- vanishing argument: the "random" polynomial.  In the reference's own proof it is the CONSTANT 1: the commitment in
    // [REF aggregator/data/batch-task.json: chunk_proofs[0]] is g[0] = (1, 2) and its evaluation is 1 (tests/test_reference_chunk_proof.py)
    // -- Scroll's halo2 fork commits no blinding polynomial; upstream PSE halo2 draws n uniform coefficients.  The verifier accepts
    // either (it only opens the commitment); ZK_VANISHING_ONE (default) does what the reference's prover did, ZK_VANISHING_UNIFORM
    // what upstream does.
    DevBuf random_coeff;
    {
        if (!random_coeff.alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
        G1Affine com;
        if (pr->vanishing_random == ZK_VANISHING_ONE) {
            ZK_HIP(ctx, hipMemsetAsync(random_coeff.p, 0, n * 32, ctx->stream));
            ZK_HIP(ctx, hipMemcpyAsync(random_coeff.p, &one, 32, hipMemcpyHostToDevice, ctx->stream));
            PK_TRY(zk_d2h(ctx, &com, srs->g, sizeof com));                    // commit(1) = g[0]
        } else {
            // n uniform coefficients: ChaCha20 in counter mode on the device, keyed from the session RNG
            uint32_t key[8];
            for (uint32_t& w_ : key) w_ = rng.next_u32();
            PK_TRY(zk_fr_random(ctx, (const uint8_t*)key, 0, 0, random_coeff.p, n));
            PK_TRY(commit_coeff(ctx, srs, random_coeff.fr(), n, &com));
        }
        tr.write_point(com);
    }
    trace.mark("random poly");
    lag.y = tr.squeeze();

    // ---- coefficient forms of everything the quotient reads and the proof opens
    std::vector<DevBuf> pz_coeff(pk->C), m_coeff(pk->L), phi_coeff(pk->L);
    {   // one batch: several columns share a launch (ntt_run_many)
        std::vector<Fr*> dsts;
        std::vector<const Fr*> srcs;
        auto want = [&](const DevBuf& lagv, DevBuf* co) -> int {
            if (!co->alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
            dsts.push_back(co->fr());
            srcs.push_back(lagv.fr());
            return ZK_OK;
        };
        for (uint32_t i = 0; i < pk->A; ++i) if (!adv_coeff[i].p) PK_TRY(want(adv_lag[i], &adv_coeff[i]));
        for (uint32_t c = 0; c < pk->C; ++c) PK_TRY(want(pz_lag[c], &pz_coeff[c]));
        for (uint32_t l = 0; l < pk->L; ++l) { PK_TRY(want(lk_m[l], &m_coeff[l])); PK_TRY(want(lk_phi[l], &phi_coeff[l])); }
        const Fr omega_inv = fr_inv_host(fr_root_of_unity(pk->k)), ninv = fr_inv_host(fr_from_u64(1ull << pk->k));
        PK_TRY(ntt_run_many(ctx, dsts.data(), srcs.data(), dsts.size(), pk->k, omega_inv, &ninv, nullptr, nullptr, false));
    }
    trace.mark("coefficient forms");
    // ---- the quotient's constraints in halo2's order: gates, permutation, lookups (folded with y below)
    const int32_t rot_last = -(int32_t)(pk->bf + 1);
    // The extended domain is evaluated one coset at a time (g_r = zeta * omega_ext^r, r < 2^(ext_k-k)):
    // every column the program reads is taken to that coset with a size-n transform of its
    // coefficients, the program runs over n rows (rotations are index shifts inside a coset), and
    // the result, divided by the vanishing polynomial -- a constant g_r^n - 1 on a coset of H --
    // lands at stride 2^(ext_k-k) in the extended buffer.  Live memory is one n-row block per
    // column instead of 2^(ext_k-k) of them: what lets 10^3-column circuits fit.
    //
    // Degree classes.  h = (sum_i y^(K-1-i) g_i) / (X^n - 1) is linear in the constraints, and a constraint of
    // degree g only needs (g - 1) n evaluation points, i.e. the cosets r that are multiples of
    // 2^(E - e), e = ceil(log2(g - 1)), E = ext_k - k: the extended domain of size 2^(k+e) is the union of
    // exactly those cosets.  The constraints are therefore grouped by e; class e is evaluated on its 2^e
    // cosets only, brought to coefficient form over its own (smaller) extended domain and added to h.  A
    // column that occurs only in low-degree constraints needs 2^e coset transforms instead of 2^E, and the
    // low-degree part of the program runs over 2^e n rows instead of 2^E n.  The polynomial h -- and with
    // it every proof byte -- is the same as when everything is evaluated on the full extended domain
    // (what halo2's evaluate_h does); how much is saved depends on the circuit's degree profile.
    // The plan (which class evaluates what, each class's program) is a function of the key: made once, make_quotient_plan.
    std::shared_ptr<const QuotientPlan> qplan;
    {
        std::string err;
        const int rc_plan = quotient_plan(pk, &qplan, &err);
        if (rc_plan) return ctx->fail(rc_plan, "%s", err.c_str());
    }
    const uint32_t E = ext_k - k, K = qplan->K;
    const bool sharded = pr->world > 1 && pr->gather;
    struct QClass { const Prog& prog; const std::vector<uint32_t>& refs; uint32_t last; bool used; DevBuf h; };
    std::vector<QClass> qc;
    qc.reserve(E + 1);
    for (uint32_t e = 0; e <= E; ++e) qc.push_back(QClass{qplan->cls[e].prog, qplan->cls[e].refs, qplan->cls[e].last, qplan->cls[e].used, DevBuf()});
    struct Remainder { DevBuf coeff; };
    std::vector<Remainder> rems(qplan->rems.size());
    lag.ypow.resize(K + 1);
    lag.ypow[0] = host::fr_one();
    for (uint32_t g_ = 1; g_ <= K; ++g_) lag.ypow[g_] = host::fr_mul(lag.ypow[g_ - 1], lag.y);
    if (!rems.empty()) {
        Env lag_r = lag;                       // the remainders read the Lagrange forms of Z, m and phi as well
        lag_r.perm_z = &pz_lag;
        lag_r.lk_m = &lk_m;
        lag_r.lk_phi = &lk_phi;
        std::vector<Fr*> dsts;
        for (size_t j = 0; j < rems.size(); ++j) {
            const QPlanRem& rp = qplan->rems[j];
            if (!rems[j].coeff.alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
            PK_TRY(run_program(ctx, lag_r, rp.prog, rems[j].coeff.p));                  // the parts on H, folded with y
            const F4 yp = host::fr_pow(lag.y, K - 1 - rp.last);                         // ... and weighted like the constraints they belong to
            PK_TRY(zk_fr_scale(ctx, rems[j].coeff.p, &yp, n));
            dsts.push_back(rems[j].coeff.fr());
        }
        const Fr omega_inv = fr_inv_host(fr_root_of_unity(pk->k)), ninv = fr_inv_host(fr_from_u64(1ull << pk->k));
        PK_TRY(ntt_run_many(ctx, dsts.data(), nullptr, dsts.size(), pk->k, omega_inv, &ninv, nullptr, nullptr, false));
        trace.mark("  quotient: remainders of the split constraints");
    }
    const std::vector<uint32_t>& refs = qplan->refs;          // every column any class reads
    for (uint32_t e = 0; e <= E; ++e)
        if (qc[e].used || e == E) {
            if (!qc[e].h.alloc(((size_t)n << e) * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
            if (!qc[e].used) ZK_HIP(ctx, hipMemsetAsync(qc[e].h.p, 0, ((size_t)n << e) * 32, ctx->stream));
        }
    DevBuf& h = qc[E].h;
    {
        const uint32_t nparts = 1u << E;
        auto of_key = [](uint32_t ref) { const uint32_t t = ref >> 24; return t == CT_FIXED || t == CT_SIGMA || t == CT_SPECIAL; };
        if (pk->part_cache_state < 0) {       // decide once: do the key's own cosets fit the budget?
            size_t key_slots = 0;             // (column of the key, coset) pairs some class reads: what the cache will hold
            for (uint32_t ref : refs) {
                if (!of_key(ref)) continue;
                for (uint32_t r = 0; r < nparts; ++r) {
                    bool read = false;
                    for (uint32_t e = 0; e <= E && !read; ++e)
                        read = qc[e].used && (r & ((1u << (E - e)) - 1u)) == 0 && std::find(qc[e].refs.begin(), qc[e].refs.end(), ref) != qc[e].refs.end();
                    key_slots += read;
                }
            }
            // The budget is shared by every key alive on this context (a Prover keeps the chunk, compression and aggregation
            // keys resident together): the cap (ZK_PK_COSET_CACHE_GB, default 96) or a third of the device, whichever is
            // smaller, less what other keys already hold -- and it must fit what the device has free right now, counting
            // the session pool's parked blocks as free (pool_trim gives them back).
            const char* env = getenv("ZK_PK_COSET_CACHE_GB");
            double budget = (env ? atof(env) : 96.0) * (double)(1ull << 30);
            budget = std::min(budget, (double)ctx->prop.totalGlobalMem / 3.0) - (double)ctx->coset_cache_bytes;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
            const double need = (double)key_slots * n * 32.0;
            const double session = (double)refs.size() * n * 32.0 + (double)((size_t)n << E) * 32.0 * 2.0;       // this proof's own coset buffers and h
            pk->part_cache_state = (need <= budget && need + session <= (double)free_b + (double)ctx->pool_bytes) ? 1 : 0;
            if (pk->part_cache_state) pk->part_cache.resize(nparts);
        }
        const bool cache_on = pk->part_cache_state >= 1;
        std::vector<DevBuf> part_buf(refs.size());       // one coset buffer per column that is transformed here: allocated on first use
        auto pre_coset = [&](uint32_t ref, uint32_t r) -> const void* {      // an advice column's coset r computed during the advice phases, if any
            if ((ref >> 24) != CT_ADVICE || r >= pr->adv_coset.size()) return nullptr;
            const uint32_t c_ = ref & 0xFFFFFFu;
            return c_ < pr->adv_coset[r].size() ? pr->adv_coset[r][c_].p : nullptr;
        };
        DevBuf hpart;
        if (!hpart.alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
        if (late_cosets) ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_aux, 0));          // the cosets computed beside the earlier stages are complete
        Env part = lag;
        const Fr w_n = fr_root_of_unity(k), w_ext = fr_root_of_unity(ext_k);
        Fr g = fr_zeta();
        // Sharded session: the unit of work is a (class, coset) pair -- class e on coset r costs the transforms of the columns
        // that class reads plus its program -- and the pairs are dealt to the ranks longest first (every rank computes the same
        // deal).  A rank keeps the finished, already divided n-row results of its pairs; afterwards they are all-gathered round
        // by round (round t = every rank's t-th pair) and interleaved into the classes' buffers on every rank.
        const Fr one_fr = Fr::one();
        std::vector<uint8_t> send, recv;
        DevBuf rtmp, gbuf;
        struct Pair { uint32_t e, r; double cost; };
        std::vector<std::vector<Pair>> deal(sharded ? pr->world : 1);       // per rank, in the order of evaluation (by coset, then class)
        std::vector<DevBuf> mine;                                           // this rank's finished pairs, in that order
        auto owner = [&](uint32_t e, uint32_t r) -> uint32_t {
            for (uint32_t q_ = 0; q_ < deal.size(); ++q_) for (const Pair& pp_ : deal[q_]) if (pp_.e == e && pp_.r == r) return q_;
            return 0;
        };
        if (sharded) {
            send.assign(n * 32, 0);
            recv.resize((size_t)pr->world * n * 32);
            if (!rtmp.alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
            std::vector<Pair> all;
            for (uint32_t e = 0; e <= E; ++e)
                if (qc[e].used)
                    for (uint32_t r = 0; r < nparts; r += 1u << (E - e)) all.push_back({e, r, (double)qc[e].refs.size() + (double)qc[e].prog.size() / 64.0});
            std::stable_sort(all.begin(), all.end(), [](const Pair& a, const Pair& b) { return a.cost > b.cost; });
            std::vector<double> load(pr->world, 0.0);
            for (const Pair& pp_ : all) {
                uint32_t best = 0;
                for (uint32_t q_ = 1; q_ < pr->world; ++q_) if (load[q_] < load[best]) best = q_;
                load[best] += pp_.cost;
                deal[best].push_back(pp_);
            }
            for (auto& d_ : deal) std::sort(d_.begin(), d_.end(), [](const Pair& a, const Pair& b) { return a.r != b.r ? a.r < b.r : a.e < b.e; });
        }
        // The cosets this rank works on, each with the classes active there (sharded: those of them this rank was dealt).
        struct CosetWork { uint32_t r = 0; Fr g; std::vector<uint32_t> active; std::unordered_map<uint32_t, const void*> part_of; };
        std::vector<std::pair<uint32_t, Fr>> rs;
        for (uint32_t r_ = 0; r_ < nparts; ++r_) {
            bool any = false;
            for (uint32_t e = 0; e <= E && !any; ++e) any = qc[e].used && (r_ & ((1u << (E - e)) - 1u)) == 0 && (!sharded || owner(e, r_) == pr->rank);
            if (any) rs.push_back({r_, g});
            g = g * w_ext;
        }
        // ZK_QUOTIENT_OVERLAP=1 (round 6, measured and left OFF): the transforms of coset j + 1 run on the auxiliary stream WHILE the class
        // programs of coset j run on the main one (into a second set of coset buffers: for an unsharded proof the odd cosets, which only
        // the top class reads).  Both kernels are bound by vector issue; what each leaves to barriers / operand loads the other did not
        // fill: EVM-style headline 2.393 s against 2.405 s, plain shape 0.989 against 0.979 (alternating A/B, profiles/r06_experiments.md).
        const char* ov_env = getenv("ZK_QUOTIENT_OVERLAP");
        const bool overlap = ov_env && atoi(ov_env) == 1 && rs.size() > 1 && ctx->ensure_aux();
        std::vector<DevBuf> part_buf2(overlap ? refs.size() : 0);
        struct EventPair { hipEvent_t e[2] = {nullptr, nullptr}; ~EventPair() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); } } ev_t;
        if (overlap) for (hipEvent_t& x : ev_t.e) ZK_HIP(ctx, hipEventCreateWithFlags(&x, hipEventDisableTiming));
        CosetWork work[2];
        // the columns the active classes read on coset w.r: taken from where they already are (computed ahead, the key's cache) or
        // transformed into `bufs` on the CURRENT stream of the context
        auto transform = [&](CosetWork& w, std::vector<DevBuf>& bufs) -> int {
            const uint32_t r_ = w.r;
            w.active.clear();
            for (uint32_t e = 0; e <= E; ++e)
                if (qc[e].used && (r_ & ((1u << (E - e)) - 1u)) == 0 && (!sharded || owner(e, r_) == pr->rank)) w.active.push_back(e);
            w.part_of.clear();
            std::vector<const void*> bat_src;                 // the coset transforms of this round go out as one batch
            std::vector<void*> bat_dst;
            std::vector<std::pair<uint32_t, DevBuf>> fresh_slots;      // cache slots being filled: published only once they hold their coset
            for (size_t i = 0; i < refs.size(); ++i) {
                bool needed = false;
                for (uint32_t e : w.active) needed |= std::find(qc[e].refs.begin(), qc[e].refs.end(), refs[i]) != qc[e].refs.end();
                if (!needed) continue;
                if (const void* pre = pre_coset(refs[i], r_)) { w.part_of[refs[i]] = pre; continue; }
                if (!(cache_on && of_key(refs[i])) && !bufs[i].p && !bufs[i].alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
                void* dst = bufs[i].p;
                DevBuf fresh;                             // a cache slot being filled: published only once it holds the coset
                const bool cached = cache_on && of_key(refs[i]);
                if (cached) {
                    auto it = pk->part_cache[r_].find(refs[i]);
                    if (it != pk->part_cache[r_].end()) { w.part_of[refs[i]] = it->second.p; continue; }   // computed by an earlier proof
                    // a slot lives as long as the key, not the session's pool.  When the device has no room for it: give the
                    // pool's parked blocks back and retry; if that fails too, freeze the cache (existing slots stay in use) and
                    // compute this coset into a session buffer like any witness column's -- the proof goes on, uncached.
                    const char* env_fail = getenv("ZK_PK_COSET_CACHE_FAIL_AFTER");        // test knob: the device "runs out" after this many slots
                    const bool inject = env_fail && pk->part_cache_bytes / (n * 32) + fresh_slots.size() >= (size_t)atoll(env_fail);
                    bool got = pk->part_cache_state == 1 && !inject && fresh.alloc_unpooled(n * 32);
                    if (!got && pk->part_cache_state == 1) {
                        (void)hipGetLastError();
                        ctx->pool_trim();
                        got = !inject && fresh.alloc_unpooled(n * 32);
                        if (!got) { (void)hipGetLastError(); pk->part_cache_state = 2; }
                    }
                    if (got) dst = fresh.p;
                    else {
                        if (!bufs[i].p && !bufs[i].alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
                        dst = bufs[i].p;
                    }
                }
                w.part_of[refs[i]] = dst;
                if (refs[i] == colref(CT_SPECIAL, SP_X)) PK_TRY(zk_fr_powers(ctx, &w_n, &w.g, dst, n));   // X on the coset: g * omega^i
                else {
                    const Fr* cf = (refs[i] >> 24) == CT_SPLIT_R ? ((refs[i] & 0xFFFFFFu) < rems.size() ? rems[refs[i] & 0xFFFFFFu].coeff.fr() : nullptr)
                                                                 : coeff_of(pk, refs[i], adv_coeff, inst_coeff, pz_coeff, m_coeff, phi_coeff);
                    if (!cf) return ctx->fail(ZK_ERR_INVALID_ARG, "prover: unresolved column reference 0x%08x", refs[i]);
                    bat_src.push_back(cf);
                    bat_dst.push_back(dst);
                }
                if (cached && fresh.p) fresh_slots.emplace_back(refs[i], std::move(fresh));
            }
            PK_TRY(zk_coeff_to_coset_batch(ctx, bat_src.data(), k, &w.g, bat_dst.data(), bat_src.size()));
            for (auto& fs : fresh_slots) {
                pk->part_cache_bytes += n * 32;
                ctx->coset_cache_bytes += n * 32;
                pk->part_cache[r_][fs.first] = std::move(fs.second);
            }
            return ZK_OK;
        };
        auto programs = [&](CosetWork& w) -> int {
            const uint32_t r_ = w.r;
            part.part = &w.part_of;
            Fr gn = w.g;
            for (uint32_t i = 0; i < k; ++i) gn = sqr(gn);
            const Fr vinv = fr_inv_host(gn - Fr::one());
            for (uint32_t e : w.active) {
                ctx->prof_tag = "quotient_coset";
                const int rc_q = run_program(ctx, part, qc[e].prog, hpart.p);
                ctx->prof_tag = nullptr;
                PK_TRY(rc_q);
                // class e lags (K - 1 - last) positions behind the end of the constraint list: its sum still takes y^(K-1-last)
                const F4 yp = host::fr_pow(lag.y, K - 1 - qc[e].last);
                Fr scale;
                memcpy((void*)&scale, yp.l, 32);
                scale = scale * vinv;
                if (sharded) {                 // peers receive the finished values: keep them until the exchange
                    PK_TRY(zk_fr_scale(ctx, hpart.p, &scale, n));
                    DevBuf keep;
                    if (!keep.alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
                    ZK_HIP(ctx, hipMemcpyAsync(keep.p, hpart.p, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
                    mine.push_back(std::move(keep));
                }
                PK_TRY(zk_fr_scatter_scaled(ctx, hpart.p, n, sharded ? &one_fr : &scale, qc[e].h.p, (size_t)1 << e, r_ >> (E - e)));
            }
            return ZK_OK;
        };
        for (size_t j = 0; j < rs.size(); ++j) {
            CosetWork& cur = work[j & 1];
            if (j == 0) {
                cur.r = rs[0].first; cur.g = rs[0].second;
                PK_TRY(transform(cur, part_buf));
                trace.mark("  quotient: cosets of the columns");
            }
            if (overlap && j + 1 < rs.size()) {
                // the auxiliary stream starts behind everything the main stream holds so far -- the programs of coset j - 1, which read the
                // buffer set the transforms below write; pooled blocks handed out now were last used there as well
                CosetWork& nxt = work[(j + 1) & 1];
                nxt.r = rs[j + 1].first; nxt.g = rs[j + 1].second;
                ZK_HIP(ctx, hipEventRecord(ctx->ev_aux, ctx->stream));
                ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream_aux, ctx->ev_aux, 0));
                hipStream_t main_stream = ctx->stream;
                ctx->stream = ctx->stream_aux;
                const int rc_t = transform(nxt, ((j + 1) & 1) ? part_buf2 : part_buf);
                hipError_t e_rec = rc_t ? hipSuccess : hipEventRecord(ev_t.e[(j + 1) & 1], ctx->stream_aux);
                ctx->stream = main_stream;
                PK_TRY(rc_t);
                ZK_HIP(ctx, e_rec);
            }
            if (overlap && j > 0) ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ev_t.e[j & 1], 0));
            PK_TRY(programs(cur));
            trace.mark("  quotient: program");
            if (!overlap && j + 1 < rs.size()) {
                CosetWork& nxt = work[(j + 1) & 1];
                nxt.r = rs[j + 1].first; nxt.g = rs[j + 1].second;
                PK_TRY(transform(nxt, part_buf));
                trace.mark("  quotient: cosets of the columns");
            }
        }
        if (sharded) {
            if (mine.size() != deal[pr->rank].size()) return ctx->fail(ZK_ERR_INVALID_ARG, "sharded session: %zu pairs evaluated, %zu dealt", mine.size(), deal[pr->rank].size());
            size_t rounds = 0;
            for (const auto& d_ : deal) rounds = std::max(rounds, d_.size());
            for (size_t t = 0; t < rounds; ++t) {
                const void* mine_t = t < mine.size() ? mine[t].p : hpart.p;           // a rank without a t-th pair sends filler nobody reads
                const char* got = nullptr;
                if (pr->use_comm) {
                    // in-library RCCL: the finished pairs go device to device, stream-ordered (no host copy, no synchronisation)
                    if (!gbuf.p && !gbuf.alloc((size_t)pr->world * n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
                    PK_TRY(comm_allgather_dev(ctx, mine_t, n * 32, gbuf.p));
                    got = (const char*)gbuf.p;
                } else if (pr->gather_dev) {
                    // a caller-supplied device all-gather (zk_proof_set_device_gather): device to device as well; the callback completes on return
                    if (!gbuf.p && !gbuf.alloc((size_t)pr->world * n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
                    PK_TRY(zk_ctx_sync(ctx));
                    if (pr->gather_dev(pr->gather_dev_user, mine_t, n * 32, gbuf.p)) return ctx->fail(ZK_ERR_INVALID_ARG, "sharded session: device all-gather callback failed");
                    got = (const char*)gbuf.p;
                } else {
                    if (t < mine.size()) PK_TRY(zk_d2h(ctx, send.data(), mine_t, n * 32));
                    if (pr->gather(pr->gather_user, send.data(), n * 32, recv.data())) return ctx->fail(ZK_ERR_INVALID_ARG, "sharded session: all-gather callback failed");
                }
                for (uint32_t q_ = 0; q_ < pr->world; ++q_) {
                    if (q_ == pr->rank || t >= deal[q_].size()) continue;
                    const Pair& pp_ = deal[q_][t];
                    const void* src = got ? (const void*)(got + (size_t)q_ * n * 32) : nullptr;
                    if (!src) { PK_TRY(zk_h2d(ctx, rtmp.p, recv.data() + (size_t)q_ * n * 32, n * 32)); src = rtmp.p; }
                    PK_TRY(zk_fr_scatter_scaled(ctx, src, n, &one_fr, qc[pp_.e].h.p, (size_t)1 << pp_.e, pp_.r >> (E - pp_.e)));
                }
            }
        }
    }
    // every class back to coefficients over its own extended domain; the smaller ones are added into h
    for (uint32_t e = 0; e < E; ++e) {
        if (!qc[e].used) continue;
        PK_TRY(zk_extended_to_coeff(ctx, qc[e].h.p, k + e));
    }
    PK_TRY(zk_extended_to_coeff(ctx, h.p, ext_k));
    for (uint32_t e = 0; e < E; ++e) {
        if (!qc[e].used) continue;
        PK_TRY(zk_field_vec_op(ctx, ZK_FIELD_FR, ZK_OP_ADD, h.p, qc[e].h.p, h.p, n << e));
        qc[e].h.release();
    }
    trace.mark("quotient eval + ifft");
    const uint32_t pieces = pk->d - 1;
    {
        std::vector<const void*> hptrs(pieces);
        for (uint32_t i = 0; i < pieces; ++i) hptrs[i] = h.fr() + (size_t)i * n;
        std::vector<G1Affine> coms(pieces);
        PK_TRY(sharded_commit(ctx, pr.get(), srs, 0, hptrs.data(), pieces, n, coms.data()));
        for (const G1Affine& com : coms) tr.write_point(com);
    }

    trace.mark("h commits");
    const F4 x = tr.squeeze();
    // ---- evaluations, written in halo2's order: advice queries, fixed queries, the random polynomial,
    // the permutation's sigma polynomials, per permutation set Z(x), Z(wx) (and Z(w^last x) for all but
    // the last set), per lookup phi(x), phi(wx), m(x)
    const F4 w = [&] { Fr t = fr_root_of_unity(k); F4 r; memcpy(r.l, &t, 32); return r; }();
    const F4 w_inv = host::fr_inv(w);
    auto rotate = [&](int32_t rot) { F4 p = x; const F4 b = rot >= 0 ? w : w_inv; for (int32_t i = 0; i < (rot >= 0 ? rot : -rot); ++i) p = host::fr_mul(p, b); return p; };
    const int64_t nn = (int64_t)n;
    auto norm_rot = [&](int32_t rot) { return (int32_t)((((int64_t)rot % nn) + nn) % nn); };    // rotations are points: x w^rot, rot mod n
    struct Open { const Fr* poly; int32_t rot; F4 eval; };
    std::vector<Open> evals;
    auto eval_at = [&](const Fr* coeffs, int32_t rot, F4* out) -> int { const F4 pt = rotate(rot); return zk_poly_eval(ctx, coeffs, n, &pt, out); };
    for (const Query& qy : pk->adv_q) evals.push_back({adv_coeff[qy.idx].fr(), qy.rot, host::fr_zero()});
    const size_t e_fix = evals.size();
    for (const Query& qy : pk->fix_q) evals.push_back({pk->fixed_coeff[qy.idx].fr(), qy.rot, host::fr_zero()});
    const size_t e_random = evals.size();
    evals.push_back({random_coeff.fr(), 0, host::fr_zero()});
    const size_t e_sigma = evals.size();
    for (uint32_t j = 0; j < pk->P; ++j) evals.push_back({pk->sigma_coeff[j].fr(), 0, host::fr_zero()});
    std::vector<size_t> e_pz(pk->C);
    for (uint32_t c = 0; c < pk->C; ++c) {
        e_pz[c] = evals.size();
        for (int32_t rot : {0, 1}) evals.push_back({pz_coeff[c].fr(), rot, host::fr_zero()});
        if (c + 1 < pk->C) evals.push_back({pz_coeff[c].fr(), rot_last, host::fr_zero()});
    }
    const size_t e_lk = evals.size();
    for (uint32_t l = 0; l < pk->L; ++l) {
        for (int32_t rot : {0, 1}) evals.push_back({phi_coeff[l].fr(), rot, host::fr_zero()});
        evals.push_back({m_coeff[l].fr(), 0, host::fr_zero()});
    }
    {   // every (polynomial, point) pair in one pass: one table per distinct point, one Horner launch, one download
        std::vector<int32_t> distinct;
        std::vector<F4> points;
        std::vector<uint32_t> pidx(evals.size());
        std::vector<const void*> ptrs(evals.size());
        for (size_t i = 0; i < evals.size(); ++i) {
            const int32_t nr = norm_rot(evals[i].rot);
            size_t at = std::find(distinct.begin(), distinct.end(), nr) - distinct.begin();
            if (at == distinct.size()) { distinct.push_back(nr); points.push_back(rotate(evals[i].rot)); }
            pidx[i] = (uint32_t)at;
            ptrs[i] = evals[i].poly;
        }
        std::vector<F4> vals(evals.size());
        if (sharded) {
            // sharded session: pair i is evaluated by rank i mod world; the 32-byte values are all-gathered (every rank holds every
            // coefficient form, so which rank evaluates what is free to choose)
            const size_t per = (evals.size() + pr->world - 1) / pr->world;
            std::vector<const void*> my_ptrs;
            std::vector<uint32_t> my_pidx;
            for (size_t i = pr->rank; i < evals.size(); i += pr->world) { my_ptrs.push_back(ptrs[i]); my_pidx.push_back(pidx[i]); }
            std::vector<F4> local(per, host::fr_zero()), all(per * pr->world);
            if (!my_ptrs.empty()) PK_TRY(zk_poly_eval_pairs(ctx, my_ptrs.data(), my_pidx.data(), my_ptrs.size(), points.data(), points.size(), n, local.data()));
            if (pr->gather(pr->gather_user, local.data(), per * sizeof(F4), all.data())) return ctx->fail(ZK_ERR_INVALID_ARG, "sharded session: all-gather callback failed");
            for (size_t i = 0; i < evals.size(); ++i) vals[i] = all[(i % pr->world) * per + i / pr->world];
        } else
            PK_TRY(zk_poly_eval_pairs(ctx, ptrs.data(), pidx.data(), evals.size(), points.data(), points.size(), n, vals.data()));
        for (size_t i = 0; i < evals.size(); ++i) evals[i].eval = vals[i];
        for (const Open& o : evals) tr.write_scalar(o.eval);
    }
    trace.mark("evaluations");
    // h(X) = sum_i x^(n i) h_i(X): opened at x, the verifier derives its expected value itself
    DevBuf hcomb;
    F4 h_eval;
    {
        if (!hcomb.alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
        F4 xn = x;
        for (uint32_t i = 0; i < k; ++i) xn = host::fr_mul(xn, xn);
        std::vector<uint32_t> words;
        std::vector<const void*> cols;
        for (uint32_t i = pieces; i-- > 0;) { words.insert(words.end(), {Q_PUSH_COL, (uint32_t)cols.size(), 0u, Q_FOLD, 0u, 0u}); cols.push_back(h.fr() + (size_t)i * n); }
        PK_TRY(zk_quotient_eval(ctx, words.data(), (uint32_t)(words.size() / 3), cols.data(), (uint32_t)cols.size(), &xn, 1, k, k, 0, hcomb.p));
        PK_TRY(eval_at(hcomb.fr(), 0, &h_eval));
    }
    trace.mark("h recombination");
    // ---- the multi-open's queries in halo2's order (plonk::prover::create_proof): advice, permutation
    // products (x and wx per set, then w^last x for all but the last set in REVERSE set order), lookups,
    // fixed, permutation sigma, then h and the random polynomial
    std::vector<Open> queries;
    for (size_t i = 0; i < e_fix; ++i) queries.push_back(evals[i]);
    for (uint32_t c = 0; c < pk->C; ++c) { queries.push_back(evals[e_pz[c]]); queries.push_back(evals[e_pz[c] + 1]); }
    for (uint32_t c = pk->C > 1 ? pk->C - 1 : 0; c-- > 0;) queries.push_back(evals[e_pz[c] + 2]);
    for (size_t i = e_lk; i < evals.size(); ++i) queries.push_back(evals[i]);
    for (size_t i = e_fix; i < e_random; ++i) queries.push_back(evals[i]);
    for (size_t i = e_sigma; i < e_sigma + pk->P; ++i) queries.push_back(evals[i]);
    queries.push_back({hcomb.fr(), 0, h_eval});
    queries.push_back(evals[e_random]);
    for (Open& o : queries) o.rot = norm_rot(o.rot);
    auto point_of = [&](int32_t nrot) { return rotate(nrot > nn / 2 ? (int32_t)(nrot - nn) : nrot); };
    // sum_j ch^j * polys[j] on the device: Horner from the last polynomial down (FOLD multiplies the accumulator by ch)
    auto lincomb = [&](const std::vector<const void*>& polys, const F4& ch, void* d_out) -> int {
        std::vector<uint32_t> words;
        std::vector<const void*> cols;
        for (size_t j = polys.size(); j-- > 0;) { words.insert(words.end(), {Q_PUSH_COL, (uint32_t)cols.size(), 0u, Q_FOLD, 0u, 0u}); cols.push_back(polys[j]); }
        return zk_quotient_eval(ctx, words.data(), (uint32_t)(words.size() / 3), cols.data(), (uint32_t)cols.size(), &ch, 1, k, k, 0, d_out);
    };
    DevBuf batch, wit;
    if (!batch.alloc(n * 32) || !wit.alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
    if (pr->multiopen == ZK_MULTIOPEN_SHPLONK) {
        // ---- SHPLONK / BDFG21 (poly::kzg::multiopen::shplonk::ProverSHPLONK::create_proof): what
        // the reference's call sites instantiate.  Two commitments whatever the number of polynomials and points.
        const F4 y = tr.squeeze();
        // construct_intermediate_sets: polynomials in order of first appearance with their point sets;
        // rotation sets in order of first appearance, each listing its polynomials
        struct PolyQ { const Fr* poly; std::vector<int32_t> rots; std::vector<F4> evals; };
        std::vector<PolyQ> polys;
        for (const Open& o : queries) {
            auto it = std::find_if(polys.begin(), polys.end(), [&](const PolyQ& p) { return p.poly == o.poly; });
            if (it == polys.end()) { polys.push_back({o.poly, {}, {}}); it = polys.end() - 1; }
            if (std::find(it->rots.begin(), it->rots.end(), o.rot) == it->rots.end()) { it->rots.push_back(o.rot); it->evals.push_back(o.eval); }
        }
        struct Set { std::vector<int32_t> rots; std::vector<size_t> members; };
        std::vector<Set> sets;
        std::vector<int32_t> super;                  // every point that is opened somewhere
        for (size_t pi = 0; pi < polys.size(); ++pi) {
            std::vector<int32_t> key = polys[pi].rots;
            std::sort(key.begin(), key.end());
            auto it = std::find_if(sets.begin(), sets.end(), [&](const Set& s_) { return s_.rots == key; });
            if (it == sets.end()) { sets.push_back({key, {}}); it = sets.end() - 1; }
            it->members.push_back(pi);
            for (int32_t r_ : key) if (std::find(super.begin(), super.end(), r_) == super.end()) super.push_back(r_);
        }
        const F4 v = tr.squeeze();
        // r_ij(X): interpolation of polynomial j's evaluations over its set's points (degree < |S|), host side.  The Lagrange basis
        // of a set -- L_a(X) = prod_{b != a} (X - x_b) / (x_a - x_b), |S|^3 products -- is built ONCE per set; a member then costs
        // |S|^2.  (Per member it was 3.9 ms of the Keccak-shape proof: 48 columns opened at the same 14 points.)
        auto lagrange_basis = [&](const std::vector<F4>& xs) {
            const size_t m = xs.size();
            std::vector<std::vector<F4>> basis(m);
            for (size_t a = 0; a < m; ++a) {
                std::vector<F4> num{host::fr_one()};       // prod_{b != a} (X - x_b)
                F4 den = host::fr_one();
                for (size_t b2 = 0; b2 < m; ++b2) {
                    if (b2 == a) continue;
                    std::vector<F4> nx(num.size() + 1, host::fr_zero());
                    for (size_t t = 0; t < num.size(); ++t) { nx[t + 1] = host::fr_add(nx[t + 1], num[t]); nx[t] = host::fr_sub(nx[t], host::fr_mul(num[t], xs[b2])); }
                    num.swap(nx);
                    den = host::fr_mul(den, host::fr_sub(xs[a], xs[b2]));
                }
                const F4 dinv = host::fr_inv(den);
                for (F4& c : num) c = host::fr_mul(c, dinv);
                basis[a] = std::move(num);
            }
            return basis;
        };
        auto interpolate = [&](const std::vector<std::vector<F4>>& basis, const std::vector<F4>& ys) {
            const size_t m = basis.size();
            std::vector<F4> out(m, host::fr_zero());
            for (size_t a = 0; a < m; ++a)
                for (size_t t = 0; t < m; ++t) out[t] = host::fr_add(out[t], host::fr_mul(basis[a][t], ys[a]));
            return out;
        };
        auto eval_small = [&](const std::vector<F4>& c, const F4& at) { F4 acc = host::fr_zero(); for (size_t t = c.size(); t-- > 0;) acc = host::fr_add(host::fr_mul(acc, at), c[t]); return acc; };
        std::vector<DevBuf> qfull(sets.size()), hset(sets.size());
        DevBuf coset_x, coset_ginv, coset_den;                 // division of large rotation sets on a coset of the domain (below)
        const F4 coset_g = host::fr_from_u64(7);
        std::vector<std::vector<F4>> Rset(sets.size());       // R_i(X) = sum_j y^j r_ij(X)
        DevBuf tmp;
        if (!tmp.alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
        for (size_t si = 0; si < sets.size(); ++si) {
            const Set& st = sets[si];
            std::vector<F4> zs;
            for (int32_t r_ : st.rots) zs.push_back(point_of(r_));
            // N_i(X) = sum_j y^j (P_ij(X) - r_ij(X)):  Qfull_i = sum_j y^j P_ij on the device, R_i on the host
            std::vector<const void*> members;
            const std::vector<std::vector<F4>> basis = lagrange_basis(zs);
            std::vector<F4> R(st.rots.size(), host::fr_zero());
            F4 ypow = host::fr_one();
            for (size_t pi : st.members) {
                members.push_back(polys[pi].poly);
                std::vector<F4> ys(st.rots.size());
                for (size_t a = 0; a < st.rots.size(); ++a) {
                    const size_t where = std::find(polys[pi].rots.begin(), polys[pi].rots.end(), st.rots[a]) - polys[pi].rots.begin();
                    ys[a] = polys[pi].evals[where];
                }
                const std::vector<F4> rj = interpolate(basis, ys);
                for (size_t t = 0; t < R.size(); ++t) R[t] = host::fr_add(R[t], host::fr_mul(ypow, rj[t]));
                ypow = host::fr_mul(ypow, y);
            }
            Rset[si] = R;
            if (!qfull[si].alloc(n * 32) || !hset[si].alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
            trace.mark("  shplonk: interpolation (host)");
            PK_TRY(lincomb(members, y, qfull[si].p));
            trace.mark("  shplonk: set combination");
            // Q_i = N_i / prod (X - z): subtract R_i from the low coefficients, divide point by point
            PK_TRY(zk_d2d(ctx, hset[si].p, qfull[si].p, n * 32));
            std::vector<F4> low(R.size());
            PK_TRY(zk_d2h(ctx, low.data(), hset[si].p, R.size() * 32));
            for (size_t t = 0; t < R.size(); ++t) low[t] = host::fr_sub(low[t], R[t]);
            PK_TRY(zk_h2d(ctx, hset[si].p, low.data(), R.size() * 32));
            // Division by Z_S(X) = prod (X - z).  Few points: one synthetic division per point.  Many (a column opened at 14
            // rotations: 14 dependent scans, 1.7 ms of the Keccak-shape proof at k = 18): N_i has degree < n and vanishes on S,
            // so Q_i = N_i / Z_S is fixed by its values on a coset g H of the domain -- transform, divide point by point by
            // Z_S(g w^j) (one batch inversion), transform back: the cost no longer depends on |S|.
            bool by_coset = zs.size() >= 4 && n >= 4 * zs.size();
            if (by_coset) {
                // no point of S may lie on the coset (z = x w^rot for a random x: z^n = g^n has probability n / r)
                F4 gn = coset_g;
                for (uint32_t i = 0; i < k; ++i) gn = host::fr_mul(gn, gn);
                for (const F4& z : zs) { F4 zn = z; for (uint32_t i = 0; i < k; ++i) zn = host::fr_mul(zn, zn); if (host::fr_eq(zn, gn)) by_coset = false; }
            }
            if (by_coset) {
                if (!coset_x.p) {        // g w^j and g^-i, once per proof
                    if (!coset_x.alloc(n * 32) || !coset_ginv.alloc(n * 32) || !coset_den.alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "prover: alloc failed");
                    const F4 ginv = host::fr_inv(coset_g), one = host::fr_one();
                    PK_TRY(zk_fr_powers(ctx, &w, &coset_g, coset_x.p, n));
                    PK_TRY(zk_fr_powers(ctx, &ginv, &one, coset_ginv.p, n));
                }
                std::vector<uint32_t> words;
                std::vector<F4> cs;
                for (size_t t = 0; t < zs.size(); ++t) {
                    words.insert(words.end(), {Q_PUSH_COL, 0u, 0u, Q_ADD_CONST, (uint32_t)t, 0u});
                    if (t) words.insert(words.end(), {Q_MUL, 0u, 0u});
                    cs.push_back(host::fr_sub(host::fr_zero(), zs[t]));
                }
                words.insert(words.end(), {Q_FOLD, (uint32_t)zs.size(), 0u});
                cs.push_back(host::fr_one());
                const void* xcol[1] = {coset_x.p};
                PK_TRY(zk_quotient_eval(ctx, words.data(), (uint32_t)(words.size() / 3), xcol, 1, cs.data(), (uint32_t)cs.size(), k, k, 0, coset_den.p));     // Z_S(g w^j)
                PK_TRY(zk_fr_batch_invert(ctx, coset_den.p, n));
                PK_TRY(zk_coeff_to_coset(ctx, hset[si].p, k, &coset_g, tmp.p));                                                                          // N_i(g w^j)
                const uint32_t mulw[] = {Q_PUSH_COL, 0u, 0u, Q_PUSH_COL, 1u, 0u, Q_MUL, 0u, 0u, Q_FOLD, 0u, 0u};
                const F4 one = host::fr_one();
                const void* c2[2] = {tmp.p, coset_den.p};
                PK_TRY(zk_quotient_eval(ctx, mulw, 4, c2, 2, &one, 1, k, k, 0, hset[si].p));                                                              // Q_i(g w^j)
                PK_TRY(zk_ntt(ctx, hset[si].p, k, 1));                                                                                                  // q_i g^i
                const void* c3[2] = {hset[si].p, coset_ginv.p};
                PK_TRY(zk_quotient_eval(ctx, mulw, 4, c3, 2, &one, 1, k, k, 0, tmp.p));
                PK_TRY(zk_d2d(ctx, hset[si].p, tmp.p, n * 32));
            } else {
                size_t len = n;
                for (const F4& z : zs) {
                    PK_TRY(zk_kate_division(ctx, hset[si].p, len, &z, tmp.p));
                    --len;
                    PK_TRY(zk_d2d(ctx, hset[si].p, tmp.p, len * 32));
                    ZK_HIP(ctx, hipMemsetAsync((char*)hset[si].p + len * 32, 0, (n - len) * 32, ctx->stream));
                }
            }
            trace.mark("  shplonk: set division");
        }
        // h = sum_i v^i Q_i; commit
        {
            std::vector<const void*> qs;
            for (size_t si = 0; si < sets.size(); ++si) qs.push_back(hset[si].p);
            PK_TRY(lincomb(qs, v, batch.p));
            G1Affine com;
            PK_TRY(commit_coeff(ctx, srs, batch.fr(), n, &com));
            tr.write_point(com);
        }
        trace.mark("  shplonk: h");
        const F4 u = tr.squeeze();
        // L(X) = sum_i v^i Z_{T \ S_i}(u) (Qfull_i(X) - R_i(u)) - Z_T(u) h(X);  the proof's second point commits to
        // L(X) / (X - u), normalised by 1 / Z_{T \ S_0}(u) (the verifier scales the first set's term to one)
        std::vector<F4> coef(sets.size() + 1);
        F4 zT = host::fr_one(), constant = host::fr_zero(), vpow = host::fr_one(), z0_inv = host::fr_one();
        for (int32_t r_ : super) zT = host::fr_mul(zT, host::fr_sub(u, point_of(r_)));
        for (size_t si = 0; si < sets.size(); ++si) {
            F4 zdiff = host::fr_one();
            for (int32_t r_ : super) if (std::find(sets[si].rots.begin(), sets[si].rots.end(), r_) == sets[si].rots.end()) zdiff = host::fr_mul(zdiff, host::fr_sub(u, point_of(r_)));
            if (si == 0) z0_inv = host::fr_inv(zdiff);
            coef[si] = host::fr_mul(host::fr_mul(vpow, zdiff), z0_inv);
            constant = host::fr_add(constant, host::fr_mul(coef[si], eval_small(Rset[si], u)));
            vpow = host::fr_mul(vpow, v);
        }
        coef[sets.size()] = host::fr_mul(zT, z0_inv);
        {
            std::vector<uint32_t> words;
            std::vector<const void*> cols;
            for (size_t si = 0; si < sets.size(); ++si) {
                words.insert(words.end(), {Q_PUSH_COL, (uint32_t)cols.size(), 0u, Q_MUL_CONST, (uint32_t)si, 0u});
                if (si) words.insert(words.end(), {Q_ADD, 0u, 0u});
                cols.push_back(qfull[si].p);
            }
            words.insert(words.end(), {Q_PUSH_COL, (uint32_t)cols.size(), 0u, Q_MUL_CONST, (uint32_t)sets.size(), 0u, Q_SUB, 0u, 0u});
            cols.push_back(batch.p);
            coef.push_back(host::fr_one());
            words.insert(words.end(), {Q_FOLD, (uint32_t)sets.size() + 1, 0u});
            PK_TRY(zk_quotient_eval(ctx, words.data(), (uint32_t)(words.size() / 3), cols.data(), (uint32_t)cols.size(), coef.data(), (uint32_t)coef.size(), k, k, 0, tmp.p));
            F4 c0;
            PK_TRY(zk_d2h(ctx, &c0, tmp.p, 32));
            c0 = host::fr_sub(c0, constant);
            PK_TRY(zk_h2d(ctx, tmp.p, &c0, 32));
            PK_TRY(zk_kate_division(ctx, tmp.p, n, &u, wit.p));
            G1Affine com;
            PK_TRY(commit_coeff(ctx, srs, wit.fr(), n - 1, &com));
            tr.write_point(com);
        }
        PK_TRY(zk_ctx_sync(ctx));
        trace.mark("multiopen (shplonk)");
        if (tr.err) return ctx->fail(ZK_ERR_INVALID_ARG, "transcript failed with status %d (external callback, or the identity point in a Poseidon / EVM transcript)", tr.err);
        *proof_len = tr.proof.size();
        if (tr.proof.size() > proof_cap) return ctx->fail(ZK_ERR_INVALID_ARG, "proof buffer too small: need %zu bytes", tr.proof.size());
        memcpy(h_proof, tr.proof.data(), tr.proof.size());
        return ZK_OK;
    }
    // ---- GWC multi-open (poly::kzg::multiopen::gwc::ProverGWC::create_proof): one witness per distinct
    // point in order of first appearance, the point's polynomials combined with ascending powers of v
    const F4 v = tr.squeeze();
    std::vector<int32_t> rots;
    for (const Open& o : queries) if (std::find(rots.begin(), rots.end(), o.rot) == rots.end()) rots.push_back(o.rot);
    for (int32_t rot : rots) {
        std::vector<const void*> members;
        for (const Open& o : queries) if (o.rot == rot) members.push_back(o.poly);
        PK_TRY(lincomb(members, v, batch.p));
        const F4 z = point_of(rot);
        PK_TRY(zk_kate_division(ctx, batch.p, n, &z, wit.p));
        G1Affine com;
        PK_TRY(commit_coeff(ctx, srs, wit.fr(), n - 1, &com));
        tr.write_point(com);
    }
    PK_TRY(zk_ctx_sync(ctx));
    trace.mark("multiopen");
    if (tr.err) return ctx->fail(ZK_ERR_INVALID_ARG, "transcript failed with status %d (external callback, or the identity point in a Poseidon / EVM transcript)", tr.err);
    *proof_len = tr.proof.size();
    if (tr.proof.size() > proof_cap) return ctx->fail(ZK_ERR_INVALID_ARG, "proof buffer too small: need %zu bytes", tr.proof.size());
    memcpy(h_proof, tr.proof.data(), tr.proof.size());
    return ZK_OK;
}

// ---- host-only transcript objects and hashes (no context, no device) -----------------------------
struct zk_transcript { host::Transcript tr; };
zk_transcript* zk_transcript_new(int kind) {
    if (kind != ZK_TRANSCRIPT_BLAKE2B && kind != ZK_TRANSCRIPT_POSEIDON && kind != ZK_TRANSCRIPT_EVM) return nullptr;
    zk_transcript* t = new (std::nothrow) zk_transcript();
    if (t) t->tr.reset(kind);
    return t;
}
void zk_transcript_free(zk_transcript* t) { delete t; }
static int transcript_status(zk_transcript* t) { const int e = t->tr.err; t->tr.err = 0; return e; }
int zk_transcript_common_point(zk_transcript* t, const void* affine64) {
    if (!t || !affine64) return ZK_ERR_INVALID_ARG;
    G1Affine p; memcpy((void*)&p, affine64, 64);
    t->tr.common_point(p);
    return transcript_status(t);
}
int zk_transcript_common_scalar(zk_transcript* t, const void* fr32) {
    if (!t || !fr32) return ZK_ERR_INVALID_ARG;
    F4 s_; memcpy(s_.l, fr32, 32);
    t->tr.common_scalar(s_);
    return transcript_status(t);
}
int zk_transcript_write_point(zk_transcript* t, const void* affine64) {
    if (!t || !affine64) return ZK_ERR_INVALID_ARG;
    G1Affine p; memcpy((void*)&p, affine64, 64);
    if (t->tr.kind != ZK_TRANSCRIPT_BLAKE2B && p.is_identity()) return ZK_ERR_INVALID_ARG;
    t->tr.write_point(p);
    return transcript_status(t);
}
int zk_transcript_write_scalar(zk_transcript* t, const void* fr32) {
    if (!t || !fr32) return ZK_ERR_INVALID_ARG;
    F4 s_; memcpy(s_.l, fr32, 32);
    t->tr.write_scalar(s_);
    return transcript_status(t);
}
int zk_transcript_squeeze(zk_transcript* t, void* fr32_out) {
    if (!t || !fr32_out) return ZK_ERR_INVALID_ARG;
    const F4 c = t->tr.squeeze();
    memcpy(fr32_out, c.l, 32);
    return transcript_status(t);
}
size_t zk_transcript_proof(const zk_transcript* t, const void** data) {
    if (!t) return 0;
    if (data) *data = t->tr.proof.data();
    return t->tr.proof.size();
}
int zk_host_keccak256(const void* data, size_t len, void* out32) {
    if ((!data && len) || !out32) return ZK_ERR_INVALID_ARG;
    host::keccak256((const uint8_t*)data, len, (uint8_t*)out32);
    return ZK_OK;
}
int zk_host_poseidon_permute(void* state5_fr32) {
    if (!state5_fr32) return ZK_ERR_INVALID_ARG;
    F4 st[host::PoseidonSpec::T];
    memcpy(st, state5_fr32, sizeof st);
    host::PoseidonSpec::get().permute(st);
    memcpy(state5_fr32, st, sizeof st);
    return ZK_OK;
}
int zk_host_poseidon_permute_width3(void* state3_fr32) {
    if (!state3_fr32) return ZK_ERR_INVALID_ARG;
    F4 st[host::PoseidonWidth3::T];
    memcpy(st, state3_fr32, sizeof st);
    host::PoseidonWidth3::get().permute(st);
    memcpy(state3_fr32, st, sizeof st);
    return ZK_OK;
}

// ---- dev::MockProver restated for the device ------------------------------------------------------------------
// halo2 `MockProver::run(k, &circuit, instances)` + `verify_par / verify_at_rows_par` [REF zkevm-circuits/src/test_util.rs:272],
// [REF prover/src/common/prover/mock.rs:18-19]: no commitments, no transcript -- every gate polynomial must vanish on the selected
// usable rows, every lookup input must occur among the usable rows of its table, every cell of a permutation column must equal
// the cell sigma maps it to.  What upstream derives from its region bookkeeping (CellNotAssigned, ConstraintPoisoned) has no
// counterpart here: the witness arrives as finished columns.
//   gates: one pass over all constraints folded with a random y finds out WHETHER any selected row fails (a satisfied witness
//     costs that one pass); only then one pass per constraint lists them (exact: no randomness in what is reported);
//   lookups: theta-compressed table and inputs (random theta), the hash join of the multiplicity computation with a probe that
//     reports instead of counting;
//   copies: sigma inverted with the same hash over the identity values delta^j w^i -- the key carries no cell mapping.
int zk_host_mock_challenges(uint32_t count, void* out_fr32) {
    if (count && !out_fr32) return ZK_ERR_INVALID_ARG;
    // halo2 dev.rs: hash = blake2b("Halo2-MockProver"); challenge_i = Fr::from_uniform_bytes(hash = blake2b(hash))   (golden G3: the third one)
    uint8_t h[64];
    host::Blake2b b;
    b.init(nullptr);
    b.update("Halo2-MockProver", 16);
    b.finalize(h);
    for (uint32_t i = 0; i < count; ++i) {
        host::Blake2b c;
        c.init(nullptr);
        c.update(h, 64);
        c.finalize(h);
        const F4 v = host::fr_from_uniform(h);
        memcpy((uint8_t*)out_fr32 + (size_t)i * 32, &v, 32);
    }
    return ZK_OK;
}

// y and theta of one call: fresh randomness, nothing a witness could have been fitted to
static void mock_randomise(Env& lag) {
    std::random_device rd;
    uint8_t b[128];
    for (size_t i = 0; i < sizeof b; i += 4) { const uint32_t v = rd(); memcpy(b + i, &v, 4); }
    lag.y = host::fr_from_uniform(b);
    lag.theta = host::fr_from_uniform(b + 64);
    lag.beta = lag.gamma = host::fr_zero();
}
// the checks themselves, over columns that are on the device already (lag: advice, instance, challenges, theta and y set)
static int mock_verify_core(zk_ctx* ctx, const zk_pk* pk, const Env& lag, const uint32_t* gate_rows, size_t num_gate_rows, const uint32_t* lookup_rows, size_t num_lookup_rows,
                            zk_mock_failure* out, size_t cap, size_t* count) {
    const size_t n = (size_t)1 << pk->k;
    // upstream panics on a row id outside the usable rows ("invalid gate row id")
    for (size_t i = 0; i < num_gate_rows; ++i) if (gate_rows[i] >= pk->u) return ctx->fail(ZK_ERR_INVALID_ARG, "mock verify: gate row id %u is not a usable row (%u of them)", gate_rows[i], pk->u);
    for (size_t i = 0; i < num_lookup_rows; ++i) if (lookup_rows[i] >= pk->u) return ctx->fail(ZK_ERR_INVALID_ARG, "mock verify: lookup row id %u is not a usable row (%u of them)", lookup_rows[i], pk->u);
    // failure records and counters on the device; selected rows
    const uint32_t cap32 = (uint32_t)cap;
    DevBuf fails, ctrs, rows_g, rows_l, vals;
    if (!fails.alloc((cap ? cap : 1) * sizeof(MockFail)) || !ctrs.alloc(64) || !vals.alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "mock verify: alloc failed");
    ZK_HIP(ctx, hipMemsetAsync(ctrs.p, 0, 64, ctx->stream));
    uint32_t* d_total = (uint32_t*)ctrs.p;            // every failure
    uint32_t* d_any = d_total + 1;                    // rows the folded gate pass flags
    const uint32_t* d_rows_g = nullptr;
    const uint32_t* d_rows_l = nullptr;
    const uint32_t cnt_g = gate_rows ? (uint32_t)num_gate_rows : pk->u, cnt_l = lookup_rows ? (uint32_t)num_lookup_rows : pk->u;
    if (gate_rows && num_gate_rows) {
        if (!rows_g.alloc(num_gate_rows * 4)) return ctx->fail(ZK_ERR_OOM, "mock verify: alloc failed");
        ZK_HIP(ctx, hipMemcpyAsync(rows_g.p, gate_rows, num_gate_rows * 4, hipMemcpyHostToDevice, ctx->stream));
        d_rows_g = (const uint32_t*)rows_g.p;
    }
    if (lookup_rows && num_lookup_rows) {
        if (!rows_l.alloc(num_lookup_rows * 4)) return ctx->fail(ZK_ERR_OOM, "mock verify: alloc failed");
        ZK_HIP(ctx, hipMemcpyAsync(rows_l.p, lookup_rows, num_lookup_rows * 4, hipMemcpyHostToDevice, ctx->stream));
        d_rows_l = (const uint32_t*)rows_l.p;
    }
    MockFail* d_fails = (MockFail*)fails.p;

    // ---- gates
    if (!pk->gates.empty() && cnt_g) {
        Prog all;
        for (const Prog& g : pk->gates) { all.insert(all.end(), g.begin(), g.end()); all.push_back({Q_FOLD, C_Y, 0}); }       // in order: an intermediate is parked before it is read
        PK_TRY(run_program(ctx, lag, all, vals.p));
        PK_TRY(mock_nonzero_enqueue(ctx, vals.fr(), d_rows_g, cnt_g, 0, 0, 0, d_fails, 0, d_any));
        uint32_t any = 0;
        PK_TRY(zk_d2h(ctx, &any, d_any, 4));
        if (any) {
            // one pass per constraint; a constraint that reads intermediates other constraints parked computes them itself
            // (TmpSplit re-materialises their definitions).  A key that REUSES a slot across constraints (a later TEE_TMP of a
            // slot an earlier constraint's intermediate still depends on) cannot be re-materialised that way -- the latest
            // definition would be expanded: for such keys constraint i is evaluated by running constraints 0 .. i in order, every
            // value but the last folded away (acc = acc * 0 + value), exactly the sequence the folded pass above saw.
            const bool reuse = tmp_slots_conflict(pk->gates);
            TmpSplit tmps(1);
            for (uint32_t i = 0; i < pk->gates.size(); ++i) {
                for (auto& h : tmps.have) std::fill(h.begin(), h.end(), 0u);
                Prog one;
                if (reuse) {
                    for (uint32_t j = 0; j <= i; ++j) { one.insert(one.end(), pk->gates[j].begin(), pk->gates[j].end()); one.push_back({Q_FOLD, C_ZERO, 0}); }
                    PK_TRY(run_program(ctx, lag, one, vals.p));
                    PK_TRY(mock_nonzero_enqueue(ctx, vals.fr(), d_rows_g, cnt_g, ZK_MOCK_GATE, i, 0, d_fails, cap32, d_total));
                    continue;
                }
                tmps.append(pk->gates[i], 0, one);
                one.push_back({Q_FOLD, C_ONE, 0});
                PK_TRY(run_program(ctx, lag, one, vals.p));
                PK_TRY(mock_nonzero_enqueue(ctx, vals.fr(), d_rows_g, cnt_g, ZK_MOCK_GATE, i, 0, d_fails, cap32, d_total));
            }
            if (tmps.conflict) return ctx->fail(ZK_ERR_INVALID_ARG, "mock verify: a gate reads an intermediate no earlier gate defined");
        }
    }
    // ---- lookups
    if (pk->L && cnt_l) {
        DevBuf tab, inp;
        if (!tab.alloc(n * 32) || !inp.alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "mock verify: alloc failed");
        for (uint32_t l = 0; l < pk->L; ++l) {
            const auto& lk = pk->lookups[l];
            PB pt;
            push_compressed(pt, lk.tables); pt.fold(C_ONE);
            PK_TRY(run_program(ctx, lag, pt.g, tab.p));
            const uint32_t* slots = nullptr;
            uint32_t mask = 0;
            PK_TRY(mock_hash_build(ctx, tab.fr(), pk->u, SC_TMP, &slots, &mask));
            for (size_t a = 0; a < lk.inputs.size(); ++a) {
                PB pf;
                push_compressed(pf, lk.inputs[a]); pf.fold(C_ONE);
                PK_TRY(run_program(ctx, lag, pf.g, inp.p));
                PK_TRY(mock_probe_enqueue(ctx, inp.fr(), tab.fr(), slots, mask, d_rows_l, cnt_l, ZK_MOCK_LOOKUP, l, (uint32_t)a, d_fails, cap32, d_total));
            }
        }
    }
    // ---- copy constraints (all rows: upstream walks the whole mapping, unusable rows map to themselves)
    if (pk->P) {
        const size_t cells = (size_t)pk->P << pk->k;
        if (cells > ((size_t)1 << 30)) return ctx->fail(ZK_ERR_UNSUPPORTED, "mock verify: %u permutation columns of 2^%u rows exceed the 2^30 cells one device hash holds", pk->P, pk->k);
        DevBuf ids, ptrs;
        if (!ids.alloc(cells * 32) || !ptrs.alloc((size_t)pk->P * 16)) return ctx->fail(ZK_ERR_OOM, "mock verify: alloc failed");
        const F4 omega_h = [&] { Fr w = fr_root_of_unity(pk->k); F4 o; memcpy(&o, &w, 32); return o; }();
        const F4 delta = host::fr_pow(host::fr_from_u64(7), 1ull << 28);
        F4 dj = host::fr_one();
        std::vector<const void*> hp(2 * (size_t)pk->P);
        for (uint32_t j = 0; j < pk->P; ++j) {
            PK_TRY(zk_fr_powers(ctx, &omega_h, &dj, (char*)ids.p + ((size_t)j << pk->k) * 32, n));                // delta^j w^i
            dj = host::fr_mul(dj, delta);
            hp[j] = pk->sigma_lag[j].p;
            hp[pk->P + j] = resolve_col(lag, colref(pk->perm_cols[j].first, pk->perm_cols[j].second));
            if (!hp[pk->P + j]) return ctx->fail(ZK_ERR_INVALID_ARG, "mock verify: unresolved permutation column %u", j);
        }
        ZK_HIP(ctx, hipMemcpyAsync(ptrs.p, hp.data(), hp.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        const uint32_t* slots = nullptr;
        uint32_t mask = 0;
        PK_TRY(mock_hash_build(ctx, ids.fr(), cells, SC_TMP, &slots, &mask));
        PK_TRY(mock_perm_enqueue(ctx, (const Fr* const*)ptrs.p, (const Fr* const*)ptrs.p + pk->P, ids.fr(), slots, mask, pk->P, pk->k, ZK_MOCK_PERMUTATION, d_fails, cap32, d_total));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));           // hp is stack-owned
    }
    uint32_t total = 0;
    PK_TRY(zk_d2h(ctx, &total, d_total, 4));
    const size_t got = std::min<size_t>(total, cap);
    std::vector<MockFail> h(got);
    if (got) PK_TRY(zk_d2h(ctx, h.data(), d_fails, got * sizeof(MockFail)));
    // the device appends in whatever order its waves finish: sort (kind, index, sub, row) so that equal inputs give equal outputs
    // (when more failures exist than `cap` holds, WHICH ones were kept is still up to the device; *count says so)
    std::sort(h.begin(), h.end(), [](const MockFail& a, const MockFail& b) { return std::tie(a.kind, a.index, a.sub, a.row) < std::tie(b.kind, b.index, b.sub, b.row); });
    if (got) memcpy(out, h.data(), got * sizeof(MockFail));
    *count = total;
    return ZK_OK;
}


int zk_mock_verify(zk_ctx* ctx, const zk_pk* pk, const void* const* h_advice, const void* const* h_instance, const void* h_challenges,
                   const uint32_t* gate_rows, size_t num_gate_rows, const uint32_t* lookup_rows, size_t num_lookup_rows,
                   zk_mock_failure* out, size_t cap, size_t* count) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pk && count && (out || !cap) && (h_advice || !pk->A) && (h_instance || !pk->I), "null pointer");
    ZK_REQUIRE(ctx, (gate_rows || !num_gate_rows) && (lookup_rows || !num_lookup_rows), "row list without rows");
    ZK_REQUIRE(ctx, cap <= ((size_t)1 << 24), "at most 2^24 failure records");
    static_assert(sizeof(zk_mock_failure) == sizeof(MockFail), "failure records are copied as they are");
    *count = 0;
    const size_t n = (size_t)1 << pk->k;
    PoolScope pool(ctx);
    std::vector<DevBuf> adv(pk->A), inst(pk->I);
    // the uploads read the caller's columns asynchronously: whatever way this function is left, they are through before the device
    // buffers above go back to the pool (declared after them: destroyed first) and before the caller gets its columns back
    struct DrainOnExit { zk_ctx* c; ~DrainOnExit() { (void)hipStreamSynchronize(c->stream); } } drain{ctx};
    for (uint32_t c = 0; c < pk->A; ++c) {
        ZK_REQUIRE(ctx, h_advice[c], "null advice column");
        if (!adv[c].alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "mock verify: alloc failed");
        ZK_HIP(ctx, hipMemcpyAsync(adv[c].p, h_advice[c], n * 32, hipMemcpyHostToDevice, ctx->stream));
    }
    for (uint32_t c = 0; c < pk->I; ++c) {
        ZK_REQUIRE(ctx, h_instance[c], "null instance column");
        if (!inst[c].alloc(n * 32)) return ctx->fail(ZK_ERR_OOM, "mock verify: alloc failed");
        ZK_HIP(ctx, hipMemcpyAsync(inst[c].p, h_instance[c], n * 32, hipMemcpyHostToDevice, ctx->stream));
    }
    Env lag{};
    lag.pk = pk;
    lag.advice = &adv;
    lag.instance = &inst;
    lag.challenges.resize(pk->chal_phase.size());
    if (h_challenges) memcpy(lag.challenges.data(), h_challenges, lag.challenges.size() * 32);
    else PK_TRY(zk_host_mock_challenges((uint32_t)lag.challenges.size(), lag.challenges.data()));
    mock_randomise(lag);
    return mock_verify_core(ctx, pk, lag, gate_rows, num_gate_rows, lookup_rows, num_lookup_rows, out, cap, count);
}

// The same checks inside a proving session, after its last advice phase: the columns the session holds on the device, the
// challenges its transcript produced.  What a failed verify_proof leaves open -- WHICH constraint the witness breaks -- without
// a second upload.
int zk_proof_mock_verify(zk_ctx* ctx, zk_proof* pr, const uint32_t* gate_rows, size_t num_gate_rows, const uint32_t* lookup_rows, size_t num_lookup_rows,
                         zk_mock_failure* out, size_t cap, size_t* count) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pr && count && (out || !cap), "null pointer");
    ZK_REQUIRE(ctx, (gate_rows || !num_gate_rows) && (lookup_rows || !num_lookup_rows), "row list without rows");
    ZK_REQUIRE(ctx, cap <= ((size_t)1 << 24), "at most 2^24 failure records");
    const zk_pk* pk = pr->pk;
    *count = 0;
    if (pr->phase < pk->num_phases) return ctx->fail(ZK_ERR_INVALID_ARG, "mock verify: %u of the session's %u advice phases are committed", pr->phase, pk->num_phases);
    PoolScope pool(ctx);
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));           // uploads and transforms of the last phase (copy / auxiliary streams are joined by the phase itself)
    if (pr->lag_partial) return ctx->fail(ZK_ERR_UNSUPPORTED, "mock verify: this rank of a sharded session holds some advice columns in coefficient form only (ZK_SHARD_COEFF=0 ships Lagrange values)");
    Env lag{};
    lag.pk = pk;
    lag.advice = &pr->adv_lag;
    lag.instance = &pr->inst_lag;
    lag.challenges = pr->challenges;
    mock_randomise(lag);
    return mock_verify_core(ctx, pk, lag, gate_rows, num_gate_rows, lookup_rows, num_lookup_rows, out, cap, count);
}

// One-shot create_proof: every advice column is known up front (no column depends on a challenge,
// or the caller derived them already); runs the phases back to back.
int zk_create_proof(zk_ctx* ctx, const zk_pk* pk, const void* const* h_advice, const void* const* h_instance, const uint8_t* seed16,
                    void* h_proof, size_t proof_cap, size_t* proof_len) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, pk && seed16 && h_proof && proof_len && (h_advice || !pk->A) && (h_instance || !pk->I), "null pointer");
    zk_proof* pr = nullptr;
    PK_TRY(zk_proof_begin(ctx, pk, h_instance, seed16, &pr));
    for (uint32_t ph = 0; ph < pk->num_phases; ++ph) {
        std::vector<uint32_t> idx;
        std::vector<const void*> cols;
        for (uint32_t c = 0; c < pk->A; ++c) if (pk->adv_phase[c] == ph) { idx.push_back(c); cols.push_back(h_advice[c]); }
        const int rc = zk_proof_advice_phase(ctx, pr, idx.data(), cols.data(), (uint32_t)idx.size(), nullptr, nullptr);
        if (rc) { zk_proof_abort(ctx, pr); return rc; }
    }
    return zk_proof_finish(ctx, pr, h_proof, proof_cap, proof_len);
}

}  // extern "C"
