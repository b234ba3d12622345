// Context object behind the C ABI (include/zkmi355.h): device, stream, cached NTT domains,
// a growable scratch arena, and the error string.  One host thread per context.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/zkmi355.h"
#include "ec.hip.hpp"
#include "ff.hip.hpp"

namespace zk {

struct NttDomain;   // ntt.hip

struct Scratch {
    void* ptr = nullptr;
    size_t cap = 0;
};

}  // namespace zk

struct zk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipDeviceProp_t prop{};
    // scratch arenas, grown on demand (never shrunk): index = purpose
    zk::Scratch scratch[19];
    // side stream + events for pipelining consecutive MSMs (msm.hip): created on first use
    hipStream_t stream2 = nullptr, stream2b = nullptr, stream2c = nullptr;
    std::vector<hipStream_t> owned_streams;      // every stream zk_ctx_create made (roles may share one; 'd' entries of the layout have no role at all)
    hipEvent_t ev_p1[3] = {nullptr, nullptr, nullptr}, ev_p2[3] = {nullptr, nullptr, nullptr};
    std::map<uint64_t, std::vector<uint64_t>> msm_graph_keys;   // the full key behind every hash in msm_graphs (compared on a hit)
    std::map<uint64_t, void*> msm_graphs;   // hipGraphExec_t per (column kind, pipeline, sizes, buffer addresses): msm_batch_merged's graph mode
    uint32_t msm_blinded_tail = 0;          // set by the prover around a commit batch: this many rows at the end of the hint-1 (small-valued) columns
                                            // hold field-sized blinding values; they are committed apart (k_msm_tails) so that the main MSM sees small values only
    bool msm_graph_broken = false;          // a capture or replay failed once: the plain launch path from then on
    hipEvent_t ev_sorted[2] = {nullptr, nullptr};   // sort-ahead of msm_batch_merged: workspace copy i holds a finished sort
    hipEvent_t ev_pipe = nullptr;        // joins the second MSM pipeline of small batches (msm_batch_merged)
    // copy stream: host -> device staging of the next column under the current MSM (api.hip)
    hipStream_t stream_copy = nullptr;
    hipEvent_t ev_copy = nullptr;
    // auxiliary compute stream: transforms of freshly uploaded columns run beside the commitment pipeline
    hipStream_t stream_aux = nullptr;
    hipEvent_t ev_aux = nullptr;
    // The auxiliary stream (transforms of freshly uploaded columns beside the commitment pipeline).  ZK_AUX_PRIORITY=1 creates it
    // with the highest stream priority -- a measurement knob, OFF by default: it did not help the sort-ahead experiment
    // (profiles/r03_sort_ahead.md) and it slows the witness uploads of the advice phase from 0.64 to 0.94 ms per 32 MiB column
    // (643 -> 940 ms for the SuperCircuit shape, tools/gpu_r3i.sh): a priority stream takes a hardware queue of its own, and which
    // streams share queues decides the upload rate (DESIGN "stream topology").
    bool ensure_aux() {
        // stream and event are a pair: whoever created the stream (zk_ctx_debug_delay can), the event exists when this returns true
        if (stream_aux) return ev_aux || hipEventCreateWithFlags(&ev_aux, hipEventDisableTiming) == hipSuccess;
        int least = 0, greatest = 0;
        const char* e = getenv("ZK_AUX_PRIORITY");
        const bool prio = e && atoi(e) == 1 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least;
        hipError_t rc = prio ? hipStreamCreateWithPriority(&stream_aux, hipStreamNonBlocking, greatest) : hipStreamCreateWithFlags(&stream_aux, hipStreamNonBlocking);
        if (rc != hipSuccess) { (void)hipGetLastError(); rc = hipStreamCreateWithFlags(&stream_aux, hipStreamNonBlocking); }
        if (rc != hipSuccess) { stream_aux = nullptr; return false; }
        if (!ev_aux && hipEventCreateWithFlags(&ev_aux, hipEventDisableTiming) != hipSuccess) return false;
        return true;
    }
    // RCCL communicator of this rank (comm.hip; opaque here so that rccl.h stays out of the other translation units)
    void* comm = nullptr;
    uint32_t comm_rank = 0, comm_world = 1;
    std::map<uint64_t, std::shared_ptr<zk::NttDomain>> domains;   // key: log_n | kind << 8
    std::map<uint64_t, void*> pow_tables;                         // cached two-level power tables of the coset generators
    std::vector<void*> pinned;   // small pinned host staging buffers
    // Device block pool for the prover's column buffers: a proof allocates and frees hundreds of
    // n x 32 B / 2^ext_k x 32 B blocks, and hipFree is a device-wide synchronisation.  Blocks are
    // recycled by exact size; every user is ordered on `stream`, so reuse needs no extra fence.
    std::map<size_t, std::vector<void*>> pool;
    size_t pool_bytes = 0, pool_cap = (size_t)96 << 30;
    size_t coset_cache_bytes = 0;        // device memory held by the coset caches of every proving key alive on this context (prover.hip)
    void* pool_get(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (!bytes) bytes = 256;
        auto it = pool.find(bytes);
        if (it != pool.end() && !it->second.empty()) { void* p = it->second.back(); it->second.pop_back(); pool_bytes -= bytes; return p; }
        void* p = nullptr;
        if (hipMalloc(&p, bytes) == hipSuccess) return p;
        (void)hipGetLastError();
        pool_trim();                                   // give cached blocks back and retry once
        if (hipMalloc(&p, bytes) == hipSuccess) return p;
        (void)hipGetLastError();
        return nullptr;
    }
    void pool_put(void* p, size_t bytes) {
        if (!p) return;
        bytes = (bytes + 255) & ~(size_t)255;
        if (!bytes) bytes = 256;
        if (pool_bytes + bytes > pool_cap) { (void)hipStreamSynchronize(stream); (void)hipFree(p); return; }
        pool[bytes].push_back(p);
        pool_bytes += bytes;
    }
    void pool_trim() {
        (void)hipStreamSynchronize(stream);
        for (auto& kv : pool) for (void* p : kv.second) (void)hipFree(p);
        pool.clear();
        pool_bytes = 0;
    }
    // per-kernel HIP-event profiling (zk_prof_*): off by default
    uint32_t msm_attr_set = 0;            // bit C: k_msm_m_scatter_staged<C> has its dynamic-LDS attribute on this device
    bool ntt_attr_set = false, quotient_attr_set = false;   // hipFuncSetAttribute (large dynamic LDS) is per device: remembered per context, not per process
    uint32_t ntt_fixed_attr = 0;                             // the same for the compile-time instances of the NTT passes: bit 2 (LOG_NP - 7) + HAS_PRE for the strided pass, bit 24 + LOG_NP - 7 for the last pass
    bool prof_on = false;
    bool prof_main_only = false;         // zk_prof_enable(ctx, 2): only the scopes of the roofline kernels (accumulation, transforms, evaluator) record events
    const char* prof_tag = nullptr;      // when set, zk_quotient_eval books its launch under this name (the prover tags the big coset programs)
    struct ProfEntry { double ms = 0; uint64_t count = 0; uint64_t bytes = 0; };   // bytes: algorithmic HBM bytes of the booked launches, where the scope states them
    std::map<std::string, ProfEntry> prof;
    struct ProfPending { const char* name; hipEvent_t a, b; uint64_t bytes; };
    std::vector<ProfPending> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    hipEvent_t prof_event() {
        if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void prof_resolve() {
        for (auto& p : prof_pending) {
            float ms = 0;
            if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                auto& e = prof[p.name];
                e.ms += ms;
                e.count += 1;
                e.bytes += p.bytes;
            }
            prof_pool.push_back(p.a);
            prof_pool.push_back(p.b);
        }
        prof_pending.clear();
    }

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
    // returns nullptr on failure (err set)
    void* get_scratch(int slot, size_t bytes) {
        zk::Scratch& s = scratch[slot];
        if (s.cap >= bytes) return s.ptr;
        if (s.ptr) { (void)hipStreamSynchronize(stream); (void)hipFree(s.ptr); s.ptr = nullptr; s.cap = 0; }
        size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&s.ptr, want);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&s.ptr, bytes); want = bytes; }
        if (e != hipSuccess) { s.ptr = nullptr; fail(ZK_ERR_OOM, "scratch alloc of %zu bytes failed: %s", bytes, hipGetErrorString(e)); return nullptr; }
        s.cap = want;
        return s.ptr;
    }
};

// RAII scope: records a HIP event pair around the enclosed launches on ctx->stream
struct ZkProfScope {
    // level 2 (what bench.py's timed region runs under): every MSM class (merged and grouped bucket launches, sorts, reductions), the NTT passes, the evaluator
    static bool is_main(const char* n) { return !strncmp(n, "msm_", 4) || !strncmp(n, "ntt_", 4) || !strncmp(n, "quotient", 8); }
    zk_ctx* c; const char* name; hipEvent_t a = nullptr; hipStream_t s; uint64_t bytes = 0;
    ZkProfScope(zk_ctx* ctx, const char* n, hipStream_t on = nullptr) : c(ctx), name(n), s(on ? on : ctx->stream) {
        if (c->prof_on && (!c->prof_main_only || is_main(n))) { a = c->prof_event(); (void)hipEventRecord(a, s); }
    }
    ~ZkProfScope() {
        if (a) { hipEvent_t b = c->prof_event(); (void)hipEventRecord(b, s); c->prof_pending.push_back({name, a, b, bytes}); }
    }
};

struct zk_srs {
    uint32_t k = 0;
    zk::G1Affine* g = nullptr;
    zk::G1Affine* g_lagrange = nullptr;
    // lazily built copies in R' = 2^261 Montgomery form (the MSM kernels' native base format)
    zk::G1Affine* g_rp = nullptr;
    zk::G1Affine* g_lagrange_rp = nullptr;
    // lazily built fixed-base window tables [W][2^k] (R' form) and the window size they were built for
    // basis index 2 = the prefix sums of the Lagrange basis (pfx[1]) used as a basis of its own: columns committed through
    // their first differences (runs.hip)
    zk::G1Affine* tab[3] = {nullptr, nullptr, nullptr};
    zk::G1Affine* tabn[3] = {nullptr, nullptr, nullptr};   // per-window tables (c <= 16) for the columns that fill few windows
    zk::G1Affine* pfx[2] = {nullptr, nullptr};    // prefix sums of a basis (R' form), for run-structured columns (runs.hip)
    int tab_c[3] = {0, 0, 0};
    zk::G1Affine* pfx_negtot[2] = {nullptr, nullptr};   // one point on the device: -(sum of pfx[b][j], j <= 2^k - 2), R' form
    zk::G1Affine* pfx_negtot_tab[2] = {nullptr, nullptr};   // its fixed-base table: [32 byte windows][256 digits] = digit * 2^(8 w) * point, R' form (runs.hip: k_fixed_mul)
};

#define ZK_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess) return (ctx)->fail(ZK_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

#define ZK_CHECK_LAUNCH(ctx)                                                                       \
    do {                                                                                           \
        hipError_t e__ = hipGetLastError();                                                        \
        if (e__ != hipSuccess) return (ctx)->fail(ZK_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

#define ZK_REQUIRE(ctx, cond, msg)                                                                 \
    do {                                                                                           \
        if (!(cond)) return (ctx)->fail(ZK_ERR_INVALID_ARG, "invalid argument: %s (%s:%d)", msg, __FILE__, __LINE__); \
    } while (0)

namespace zk {
enum ScratchSlot { SC_NTT = 0, SC_MSM_KEYS = 1, SC_MSM_BUCKETS = 2, SC_MSM_MISC = 3, SC_POLY = 4, SC_POLY2 = 5, SC_TMP = 6, SC_TMP2 = 7, SC_MSM_BUCKETS2 = 8, SC_MSM_RESULTS = 9, SC_QTMP = 10, SC_COMM = 11, SC_MSM_BUCKETS3 = 12, SC_MSM_BUCKETS4 = 13, SC_MSM_DESC = 14, SC_MSM_TAILS = 15, SC_NTT_AUX = 16, SC_QACC = 17, SC_QSLICE = 18 };

// host-side field helpers (slow path, used for constants / tables only)
Fr fr_from_u64(uint64_t v);
Fr fr_pow(Fr base, uint64_t e);
Fr fr_inv_host(const Fr& a);
Fr fr_root_of_unity(uint32_t log_n);   // ROOT_OF_UNITY^(2^(28-log_n))
Fr fr_zeta();

// internal entry points shared between translation units
int ntt_run(zk_ctx* ctx, Fr* d_data, uint32_t log_n, const Fr& omega, const Fr* scale /*nullable*/, const Fr* coset_pre /*nullable: a[i] *= g^i before*/, const Fr* coset_post /*nullable: out[i] *= g^i after*/,
            const Fr* d_src = nullptr /*nullable: read the input from here, d_data receives the result*/,
            bool fuse_pre = false /*coset_pre from a cached full power table inside the first pass (no separate sweep)*/);
// `count` transforms over one domain, NTT_BATCH columns per launch (d_srcs nullable: in place)
int ntt_run_many(zk_ctx* ctx, Fr* const* d_datas, const Fr* const* d_srcs, size_t count, uint32_t log_n, const Fr& omega, const Fr* scale, const Fr* coset_pre, const Fr* coset_post, bool fuse_pre);
int fr_scale_run(zk_ctx* ctx, Fr* d_a, const Fr& s, uint64_t n);
int msm_run(zk_ctx* ctx, const Fr* d_scalars, const G1Affine* d_bases, size_t n, G1Affine* h_out);
int msm_run_rp(zk_ctx* ctx, const Fr* d_scalars, const G1Affine* d_bases, const G1Affine* d_bases_rp, size_t n, G1Affine* h_out);
int msm_batch_rp(zk_ctx* ctx, const Fr* const* d_scalar_ptrs, size_t count, const G1Affine* d_bases, const G1Affine* d_bases_rp, size_t n, G1Affine* h_out);
int srs_bases_rp(zk_ctx* ctx, const zk_srs* srs, int basis, const G1Affine** out);
int srs_window_table(zk_ctx* ctx, const zk_srs* srs, int basis, size_t n, const G1Affine** out, size_t* stride);
// stage(user, it) makes the scalars of MSM `it` available (ordered before the main stream's next
// launches); it is called with 0 before the first MSM and with it + 1 once MSM `it` is enqueued,
// so an upload on the copy stream runs under the previous MSM.
typedef int (*MsmStageFn)(void* user, size_t it);
int msm_batch_tab(zk_ctx* ctx, const Fr* const* d_scalar_ptrs, size_t count, const G1Affine* d_bases, const G1Affine* d_bases_rp, const G1Affine* d_table, size_t tab_stride,
                  size_t n, G1Affine* h_out, MsmStageFn stage = nullptr, void* stage_user = nullptr);
int msm_batch_srs(zk_ctx* ctx, const zk_srs* srs, int basis, const Fr* const* d_scalar_ptrs, size_t count, size_t n, G1Affine* h_out, MsmStageFn stage = nullptr, void* stage_user = nullptr,
                  const uint8_t* narrow = nullptr /*per column: scalars fill few windows*/);
// runs.hip: commits the columns hinted as run-structured that do have few runs (done[i] = 1), leaves the others to the caller
int msm_runs_try(zk_ctx* ctx, const zk_srs* srs, int basis, const Fr* const* d_scalar_ptrs, size_t count, size_t n, const uint8_t* narrow, G1Affine* h_out, uint8_t* done);
int msm_diff_try(zk_ctx* ctx, const zk_srs* srs, int basis, const Fr* const* d_scalar_ptrs, size_t count, size_t n, const uint8_t* narrow, G1Affine* h_out, uint8_t* done);
int commit_batch_staged(zk_ctx* ctx, const zk_srs* srs, int basis, const void* const* d_scalar_ptrs, size_t count, size_t n, void* h_out_affine, MsmStageFn stage, void* stage_user, const uint8_t* narrow = nullptr);
void sample_narrow(const void* const* h_cols, size_t count, size_t n, uint8_t* narrow);   // host sampling of Montgomery-form columns: 1 = at most a quarter of the sampled values are >= 2^64
int sample_narrow_dev(zk_ctx* ctx, const void* const* d_cols, size_t count, size_t n, uint8_t* narrow);   // the same for columns resident on the device
int lookup_multiplicities_enqueue(zk_ctx* ctx, const Fr* const* d_inputs, size_t num_inputs, const Fr* d_table, size_t usable_rows, Fr* d_m, size_t n, uint32_t* d_status, bool reuse_hash = false);   // lookup.hip, no sync
// row checks of zk_mock_verify (lookup.hip; enqueued on the context's stream, no sync)
struct MockFail { uint32_t kind, index, sub, row; };          // == zk_mock_failure
int mock_hash_build(zk_ctx* ctx, const Fr* d_table, size_t rows, int scratch_slot, const uint32_t** slots_out, uint32_t* mask_out);
int mock_nonzero_enqueue(zk_ctx* ctx, const Fr* d_vals, const uint32_t* d_row_ids, uint32_t count, uint32_t kind, uint32_t index, uint32_t sub, MockFail* d_out, uint32_t cap, uint32_t* d_counter);
int mock_probe_enqueue(zk_ctx* ctx, const Fr* d_inputs, const Fr* d_table, const uint32_t* d_slots, uint32_t mask, const uint32_t* d_row_ids, uint32_t count,
                       uint32_t kind, uint32_t index, uint32_t sub, MockFail* d_out, uint32_t cap, uint32_t* d_counter);
int mock_perm_enqueue(zk_ctx* ctx, const Fr* const* d_sigma, const Fr* const* d_cols, const Fr* d_ids, const uint32_t* d_slots, uint32_t mask, uint32_t num_cols, uint32_t k,
                      uint32_t kind, MockFail* d_out, uint32_t cap, uint32_t* d_counter);
int fr_add_const_many(zk_ctx* ctx, const void* const* d_src, void* const* d_dst, size_t count, const void* h_k, size_t n);   // vec.hip: dst[c] = src[c] + k, any number of columns, no upload / sync
int g_to_lagrange(zk_ctx* ctx, const G1Affine* d_g, uint32_t k, G1Affine* d_out);   // ecntt.hip: inverse FFT over G1
bool comm_ready(const zk_ctx* ctx);                                                              // comm.hip: in-library RCCL collectives
int comm_allgather_dev(zk_ctx* ctx, const void* d_send, size_t bytes, void* d_recv);             // stream-ordered, no host sync
int comm_allgather_host(zk_ctx* ctx, const void* h_send, size_t bytes, void* h_recv);           // small host buffers, returns complete
int comm_alltoall_dev(zk_ctx* ctx, const void* d_send, size_t bytes_per_peer, void* d_recv);
void comm_release(zk_ctx* ctx);
int copy_stream_open(zk_ctx* ctx);      // copy stream starts after everything enqueued on the main stream so far
int copy_stream_fence(zk_ctx* ctx);     // main stream continues after everything enqueued on the copy stream so far
}  // namespace zk
