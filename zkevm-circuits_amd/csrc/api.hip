// C ABI glue (include/zkmi355.h): context, device memory, timers, and the NTT / MSM entry points.
#include <algorithm>
#include <set>

#include <thread>
#include "ctx.hpp"
#include "host_fq.hpp"
#include "host_util.hpp"

namespace zk {

// ---- host-side constants (slow 32-bit-limb path; used only to derive roots and tables) --------
Fr fr_from_u64(uint64_t v) {
    Fr a = Fr::zero();
    a.l[0] = (uint32_t)v;
    a.l[1] = (uint32_t)(v >> 32);
    return to_mont(a);
}
Fr fr_pow(Fr base, uint64_t e) { return pow_u64(base, e); }
Fr fr_inv_host(const Fr& a) { return inv(a); }
// halo2curves Fr::ROOT_OF_UNITY = 7^((r-1)/2^28), canonical value, converted once
static Fr root_of_unity_28() {
    Fr c = Fr{{0x60c37c9cu, 0xd34f1ed9u, 0xd39329c8u, 0x3215cf6du, 0x3dd31f74u, 0x98865ea9u, 0x166d18b7u, 0x03ddb9f5u}};
    return to_mont(c);
}
Fr fr_root_of_unity(uint32_t log_n) {
    Fr w = root_of_unity_28();
    for (uint32_t i = log_n; i < 28; ++i) w = sqr(w);
    return w;
}
// halo2curves Fr::ZETA (cube root of unity used as the extended-coset generator)
Fr fr_zeta() {
    Fr c = Fr{{0x36636f23u, 0xb8ca0b2du, 0xec2bc5e9u, 0xcc37a73fu, 0x3fd84104u, 0x048b6e19u, 0xe131a029u, 0x30644e72u}};
    return to_mont(c);
}

}  // namespace zk

using namespace zk;

extern "C" {

const char* zk_version(void) { return "zkmi355 0.1.0 (gfx950)"; }

int zk_ctx_create(int device, zk_ctx** out) {
    if (!out) return ZK_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return ZK_ERR_NO_DEVICE; }
    if (device < 0 || device >= count) return ZK_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return ZK_ERR_HIP;
    zk_ctx* c = new zk_ctx();
    c->device = device;
    if (hipGetDeviceProperties(&c->prop, device) != hipSuccess) { delete c; return ZK_ERR_HIP; }
    c->pool_cap = (size_t)((double)c->prop.totalGlobalMem * 0.8);    // blocks of finished proofs stay pooled up to 80 % of the device
    // Streams are created here, in one fixed order, not lazily where they are first needed: the runtime deals its
    // hardware queues (four by default) to streams in creation order, and which streams end up sharing a queue decides
    // how fast the witness uploads run beside the commitments (measured: 0.67 to 1.18 ms per 32 MiB column for the same
    // code, tools/gpu_r2ah.sh).  Layout letters: m main, c copy, a aux, 2 / b / x the three MSM side streams, d a stream
    // that is never used (skips a queue slot).  ZK_STREAM_LAYOUT overrides (measurement knob).
    const char* layout = getenv("ZK_STREAM_LAYOUT");
    if (!layout || !*layout) layout = "m";
    hipStream_t prev_st = nullptr;
    for (const char* q = layout; *q; ++q) {
        hipStream_t st = nullptr;
        const bool alias = *q >= 'A' && *q <= 'Z';           // upper case: this role shares the stream created last
        if (alias) st = prev_st;
        else {
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { delete c; return ZK_ERR_HIP; }
            c->owned_streams.push_back(st);
            prev_st = st;
        }
        bool ok = st != nullptr;
        switch (alias ? *q - 'A' + 'a' : *q) {
            case 'm': if (c->stream) ok = false; else c->stream = st; break;
            case 'c': if (c->stream_copy) ok = false; else { c->stream_copy = st; ok = hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming) == hipSuccess; } break;
            case 'a': if (c->stream_aux) ok = false; else { c->stream_aux = st; ok = hipEventCreateWithFlags(&c->ev_aux, hipEventDisableTiming) == hipSuccess; } break;
            case '2': if (c->stream2) ok = false; else c->stream2 = st; break;
            case 'b': if (c->stream2b) ok = false; else c->stream2b = st; break;
            case 'x': if (c->stream2c) ok = false; else c->stream2c = st; break;
            case 'd': break;
            default: ok = false;
        }
        if (!ok) { delete c; return ZK_ERR_INVALID_ARG; }
    }
    if (!c->stream) { delete c; return ZK_ERR_INVALID_ARG; }
    if (c->stream2) for (int i = 0; i < 3; ++i) { (void)hipEventCreateWithFlags(&c->ev_p1[i], hipEventDisableTiming); (void)hipEventCreateWithFlags(&c->ev_p2[i], hipEventDisableTiming); }
    c->own_stream = true;
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { delete c; return ZK_ERR_HIP; }
    *out = c;
    return ZK_OK;
}

void zk_ctx_destroy(zk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->domains.clear();
    for (auto& kv : ctx->pow_tables) (void)hipFree(kv.second);
    ctx->pool_trim();
    ctx->prof_resolve();
    for (auto e : ctx->prof_pool) (void)hipEventDestroy(e);
    for (auto& s : ctx->scratch) if (s.ptr) (void)hipFree(s.ptr);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    zk::comm_release(ctx);
    // roles may share a stream (ZK_STREAM_LAYOUT) and the main stream may belong to the caller: every handle goes once
    std::set<hipStream_t> streams(ctx->owned_streams.begin(), ctx->owned_streams.end());
    for (hipStream_t st : {ctx->stream2, ctx->stream2b, ctx->stream2c, ctx->stream_aux, ctx->stream_copy}) if (st) streams.insert(st);
    if (ctx->own_stream && ctx->stream) streams.insert(ctx->stream); else streams.erase(ctx->stream);
    for (hipStream_t st : streams) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (ctx->ev_aux) (void)hipEventDestroy(ctx->ev_aux);
    if (ctx->ev_copy) (void)hipEventDestroy(ctx->ev_copy);
    if (ctx->ev_pipe) (void)hipEventDestroy(ctx->ev_pipe);
    for (auto e : ctx->ev_sorted) if (e) (void)hipEventDestroy(e);
    for (auto& kv : ctx->msm_graphs) (void)hipGraphExecDestroy((hipGraphExec_t)kv.second);
    for (auto e : ctx->ev_p1) if (e) (void)hipEventDestroy(e);
    for (auto e : ctx->ev_p2) if (e) (void)hipEventDestroy(e);
    delete ctx;
}

const char* zk_last_error(const zk_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int zk_ctx_set_stream(zk_ctx* ctx, void* hip_stream) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) {
        // a role sharing the main stream keeps it alive (zk_ctx_destroy releases it); otherwise it goes now
        bool shared = false;
        for (hipStream_t st : {ctx->stream2, ctx->stream2b, ctx->stream2c, ctx->stream_aux, ctx->stream_copy}) shared |= st == ctx->stream;
        if (!shared) {
            ctx->owned_streams.erase(std::remove(ctx->owned_streams.begin(), ctx->owned_streams.end(), ctx->stream), ctx->owned_streams.end());
            (void)hipStreamDestroy(ctx->stream);
        }
        ctx->own_stream = false;
    }
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        ZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return ZK_OK;
}
// Joins EVERY stream the library runs on (main, copy, auxiliary transforms, the MSM side streams), the main stream last:
// when it returns nothing this context enqueued is still in flight, whichever stream an entry point put it on.
int zk_ctx_sync(zk_ctx* ctx) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    std::set<hipStream_t> side(ctx->owned_streams.begin(), ctx->owned_streams.end());
    for (hipStream_t st : {ctx->stream2, ctx->stream2b, ctx->stream2c, ctx->stream_aux, ctx->stream_copy}) if (st) side.insert(st);
    side.erase(ctx->stream);
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));      // work the side streams wait for is behind events recorded here
    for (hipStream_t st : side) ZK_HIP(ctx, hipStreamSynchronize(st));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}
// How many of the library's streams still have work in flight (hipStreamQuery): 0 right after zk_ctx_sync.
int zk_ctx_streams_busy(zk_ctx* ctx, int* busy) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, busy, "null pointer");
    std::set<hipStream_t> all(ctx->owned_streams.begin(), ctx->owned_streams.end());
    for (hipStream_t st : {ctx->stream, ctx->stream2, ctx->stream2b, ctx->stream2c, ctx->stream_aux, ctx->stream_copy}) if (st) all.insert(st);
    int n = 0;
    for (hipStream_t st : all) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipErrorNotReady) { (void)hipGetLastError(); ++n; }
        else if (e != hipSuccess) { (void)hipGetLastError(); return ctx->fail(ZK_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(e)); }
    }
    *busy = n;
    return ZK_OK;
}
// Diagnostic: keeps one of the library's streams busy for about `usec` microseconds (a one-wave kernel spinning on the
// device clock).  role: 0 main, 1 copy, 2 auxiliary transforms, 3..5 the MSM side streams (created on demand).
__global__ void k_debug_delay(uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int zk_ctx_debug_delay(zk_ctx* ctx, int role, uint32_t usec) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, role >= 0 && role <= 5 && usec <= 2000000u, "role 0..5, at most two seconds");
    hipStream_t* slot[6] = {&ctx->stream, &ctx->stream_copy, &ctx->stream_aux, &ctx->stream2, &ctx->stream2b, &ctx->stream2c};
    if (role == 1) { if (int rc = zk::copy_stream_open(ctx)) return rc; }              // copy / auxiliary stream: created with their events,
    else if (role == 2) { if (!ctx->ensure_aux()) return ctx->fail(ZK_ERR_HIP, "could not create the auxiliary stream"); }     // as the prover expects them
    else if (!*slot[role]) {
        ZK_HIP(ctx, hipStreamCreateWithFlags(slot[role], hipStreamNonBlocking));
        ctx->owned_streams.push_back(*slot[role]);
    }
    int rate_khz = 100000;                    // wall_clock64 ticks at the constant 100 MHz reference clock on gfx9
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, ctx->device);
    if (rate_khz <= 0) rate_khz = 100000;
    hipLaunchKernelGGL(k_debug_delay, dim3(1), dim3(64), 0, *slot[role], (uint64_t)usec * (uint64_t)rate_khz / 1000ull);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}
int zk_buf_alloc(zk_ctx* ctx, size_t bytes, void** d_ptr) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_ptr, "null out pointer");
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) { (void)hipGetLastError(); *d_ptr = nullptr; return ctx->fail(ZK_ERR_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
    return ZK_OK;
}
int zk_buf_free(zk_ctx* ctx, void* d_ptr) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    if (!d_ptr) return ZK_OK;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ZK_HIP(ctx, hipFree(d_ptr));
    return ZK_OK;
}
// Page-locked host memory: witness columns allocated here upload at PCIe speed and asynchronously
// (pageable memory is staged through bounce buffers at roughly half the rate).
int zk_host_alloc(zk_ctx* ctx, size_t bytes, void** h_ptr) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, h_ptr, "null pointer");
    *h_ptr = nullptr;
    hipError_t e = hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); return ctx->fail(ZK_ERR_OOM, "hipHostMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e)); }
    return ZK_OK;
}
int zk_host_free(zk_ctx* ctx, void* h_ptr) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    if (h_ptr) ZK_HIP(ctx, hipHostFree(h_ptr));
    return ZK_OK;
}
// Pin memory the caller already owns (a Vec<Fr> of witness values): same upload behaviour as
// zk_host_alloc memory without copying the column into it first.
int zk_host_register(zk_ctx* ctx, void* h_ptr, size_t bytes) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, h_ptr && bytes, "null pointer or empty range");
    hipError_t e = hipHostRegister(h_ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); return ctx->fail(ZK_ERR_HIP, "hipHostRegister of %zu bytes failed: %s", bytes, hipGetErrorString(e)); }
    return ZK_OK;
}
int zk_host_unregister(zk_ctx* ctx, void* h_ptr) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, h_ptr, "null pointer");
    hipError_t e = hipHostUnregister(h_ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return ctx->fail(ZK_ERR_HIP, "hipHostUnregister failed: %s", hipGetErrorString(e)); }
    return ZK_OK;
}
int zk_h2d(zk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, (d_dst && h_src) || !bytes, "null pointer");
    ZK_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}
int zk_d2h(zk_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, (h_dst && d_src) || !bytes, "null pointer");
    ZK_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}
int zk_d2d(zk_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, (d_dst && d_src) || !bytes, "null pointer");
    ZK_HIP(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return ZK_OK;
}
int zk_timer_start(zk_ctx* ctx) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return ZK_OK;
}
int zk_timer_stop_ms(zk_ctx* ctx, float* ms) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, ms, "null pointer");
    ZK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ZK_HIP(ctx, hipEventSynchronize(ctx->ev1));
    ZK_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return ZK_OK;
}
int zk_prof_enable(zk_ctx* ctx, int on) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ctx->prof_on = on != 0;
    ctx->prof_main_only = on == 2;
    return ZK_OK;
}
int zk_prof_reset(zk_ctx* ctx) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->prof_resolve();
    ctx->prof.clear();
    return ZK_OK;
}
int zk_prof_get(zk_ctx* ctx, const char* name, double* total_ms, uint64_t* count) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, name && total_ms && count, "null pointer");
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->prof_resolve();
    auto it = ctx->prof.find(name);
    *total_ms = it == ctx->prof.end() ? 0.0 : it->second.ms;
    *count = it == ctx->prof.end() ? 0 : it->second.count;
    return ZK_OK;
}
int zk_prof_get_bytes(zk_ctx* ctx, const char* name, uint64_t* bytes) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, name && bytes, "null pointer");
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->prof_resolve();
    auto it = ctx->prof.find(name);
    *bytes = it == ctx->prof.end() ? 0 : it->second.bytes;
    return ZK_OK;
}
int zk_prof_names(zk_ctx* ctx, char* buf, size_t len) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, buf && len, "null pointer");
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->prof_resolve();
    std::string s;
    for (auto& kv : ctx->prof) { if (!s.empty()) s += ";"; s += kv.first; }
    snprintf(buf, len, "%s", s.c_str());
    return ZK_OK;
}
int zk_device_info(zk_ctx* ctx, char* name, size_t len, int* cu_count, size_t* hbm_bytes) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    if (name && len) { snprintf(name, len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName); }
    if (cu_count) *cu_count = ctx->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = ctx->prop.totalGlobalMem;
    return ZK_OK;
}

// ---- NTT ---------------------------------------------------------------------------------------
int zk_ntt(zk_ctx* ctx, void* d_data, uint32_t log_n, int inverse) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_data, "null pointer");
    ZK_REQUIRE(ctx, log_n <= 28, "log_n exceeds the two-adicity of Fr (28)");
    Fr omega = fr_root_of_unity(log_n);
    if (!inverse) return ntt_run(ctx, (Fr*)d_data, log_n, omega, nullptr, nullptr, nullptr);
    Fr omega_inv = fr_inv_host(omega);
    Fr ninv = fr_inv_host(fr_from_u64(1ull << log_n));
    return ntt_run(ctx, (Fr*)d_data, log_n, omega_inv, &ninv, nullptr, nullptr);
}
// `count` columns of 2^log_n elements each, transformed in place; several columns share a launch (a 2^18 column alone
// occupies a quarter of the CUs)
int zk_ntt_batch(zk_ctx* ctx, void* const* d_datas, size_t count, uint32_t log_n, int inverse) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_datas || !count, "null pointer");
    ZK_REQUIRE(ctx, log_n <= 28, "log_n exceeds the two-adicity of Fr (28)");
    for (size_t i = 0; i < count; ++i) ZK_REQUIRE(ctx, d_datas[i], "null column pointer");
    Fr omega = fr_root_of_unity(log_n);
    if (!inverse) return ntt_run_many(ctx, (Fr* const*)d_datas, nullptr, count, log_n, omega, nullptr, nullptr, nullptr, false);
    Fr omega_inv = fr_inv_host(omega);
    Fr ninv = fr_inv_host(fr_from_u64(1ull << log_n));
    return ntt_run_many(ctx, (Fr* const*)d_datas, nullptr, count, log_n, omega_inv, &ninv, nullptr, nullptr, false);
}
int zk_ntt_omega(zk_ctx* ctx, void* d_data, uint32_t log_n, const void* h_omega) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_data && h_omega, "null pointer");
    ZK_REQUIRE(ctx, log_n <= 28, "log_n exceeds the two-adicity of Fr (28)");
    return ntt_run(ctx, (Fr*)d_data, log_n, *(const Fr*)h_omega, nullptr, nullptr, nullptr);
}
int zk_coeff_to_extended(zk_ctx* ctx, const void* d_coeffs, uint32_t k, uint32_t ext_k, void* d_out) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_coeffs && d_out && d_coeffs != d_out, "null or aliased pointer");
    ZK_REQUIRE(ctx, k <= ext_k && ext_k <= 28, "need k <= ext_k <= 28");
    const size_t n = (size_t)1 << k, ne = (size_t)1 << ext_k;
    ZK_HIP(ctx, hipMemcpyAsync(d_out, d_coeffs, n * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
    if (ne > n) ZK_HIP(ctx, hipMemsetAsync((char*)d_out + n * sizeof(Fr), 0, (ne - n) * sizeof(Fr), ctx->stream));
    // distribute_powers_zeta over the first n coefficients only, then the extended NTT
    Fr zeta = fr_zeta();
    int rc = ntt_run(ctx, (Fr*)d_out, ext_k, fr_root_of_unity(ext_k), nullptr, &zeta, nullptr);
    return rc;
}
// One coset of the extended domain: out[i] = f(g * omega^i), i < 2^k  (EvaluationDomain::
// coeff_to_extended_part in newer halo2: the extended domain is the union of the 2^(ext_k-k)
// cosets g_r = zeta * omega_ext^r of H, and a rotation is an index shift inside each of them).
int zk_coeff_to_coset(zk_ctx* ctx, const void* d_coeffs, uint32_t k, const void* h_g, void* d_out) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_coeffs && d_out && h_g, "null pointer");
    ZK_REQUIRE(ctx, k <= 28, "k exceeds the two-adicity of Fr (28)");
    return ntt_run(ctx, (Fr*)d_out, k, fr_root_of_unity(k), nullptr, (const Fr*)h_g, nullptr, d_coeffs == d_out ? nullptr : (const Fr*)d_coeffs, /*fuse_pre=*/true);
}
// zk_coeff_to_coset for `count` polynomials on the same coset (the columns one coset of the quotient reads)
int zk_coeff_to_coset_batch(zk_ctx* ctx, const void* const* d_coeffs, uint32_t k, const void* h_g, void* const* d_outs, size_t count) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, (d_coeffs && d_outs) || !count, "null pointer");
    ZK_REQUIRE(ctx, h_g, "null pointer");
    ZK_REQUIRE(ctx, k <= 28, "k exceeds the two-adicity of Fr (28)");
    for (size_t i = 0; i < count; ++i) ZK_REQUIRE(ctx, d_coeffs[i] && d_outs[i], "null column pointer");
    return ntt_run_many(ctx, (Fr* const*)d_outs, (const Fr* const*)d_coeffs, count, k, fr_root_of_unity(k), nullptr, (const Fr*)h_g, nullptr, /*fuse_pre=*/true);
}
int zk_extended_to_coeff(zk_ctx* ctx, void* d_ext, uint32_t ext_k) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_ext, "null pointer");
    ZK_REQUIRE(ctx, ext_k <= 28, "ext_k exceeds 28");
    Fr omega_inv = fr_inv_host(fr_root_of_unity(ext_k));
    Fr ninv = fr_inv_host(fr_from_u64(1ull << ext_k));
    Fr zeta_inv = sqr(fr_zeta());   // zeta^3 = 1
    return ntt_run(ctx, (Fr*)d_ext, ext_k, omega_inv, &ninv, nullptr, &zeta_inv);
}

// ---- MSM ---------------------------------------------------------------------------------------
int zk_msm_g1(zk_ctx* ctx, const void* d_scalars, const void* d_bases, size_t n, void* h_out_affine) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, h_out_affine && ((d_scalars && d_bases) || n == 0), "null pointer");
    return msm_run(ctx, (const Fr*)d_scalars, (const G1Affine*)d_bases, n, (G1Affine*)h_out_affine);
}
int zk_commit(zk_ctx* ctx, const zk_srs* srs, int basis, const void* d_scalars, size_t n, void* h_out_affine) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, srs && h_out_affine && (d_scalars || !n), "null pointer");
    ZK_REQUIRE(ctx, basis == 0 || basis == 1, "basis must be 0 (monomial) or 1 (Lagrange)");
    ZK_REQUIRE(ctx, n <= ((size_t)1 << srs->k), "polynomial longer than the SRS");
    const G1Affine* b = basis ? srs->g_lagrange : srs->g;
    ZK_REQUIRE(ctx, b, "SRS has no Lagrange basis");
    const Fr* sp = (const Fr*)d_scalars;
    return msm_batch_srs(ctx, srs, basis, &sp, 1, n, (G1Affine*)h_out_affine);
}
// Sum of n affine points on the HOST (no device, no context): combines the per-rank partial
// results of a point-sharded MSM after they were all-gathered as raw bytes (RCCL has no
// elliptic-curve reduction op, so the reduce step of the "all-reduce" runs here).
int zk_g1_sum_host(const void* h_points_affine, size_t n, void* h_out_affine) {
    if (!h_out_affine || (!h_points_affine && n)) return ZK_ERR_INVALID_ARG;
    const G1Affine* p = (const G1Affine*)h_points_affine;
    host::PXyzz acc;
    memset(&acc, 0, sizeof acc);
    const host::F4 one = host::fone<host::FqC>();
    for (size_t i = 0; i < n; ++i) {
        if (p[i].is_identity()) continue;
        host::PXyzz q;
        memcpy(&q.x, &p[i].x, 32);
        memcpy(&q.y, &p[i].y, 32);
        q.zz = one;
        q.zzz = one;
        acc = host::padd(acc, q);
    }
    host::pto_affine(acc, (G1Affine*)h_out_affine);
    return ZK_OK;
}

// Batch of commitments over the same basis: consecutive MSMs are pipelined (the latency-bound
// bucket reduction of column i runs on a side stream under the sort + accumulation of column i+1).
int zk_commit_batch(zk_ctx* ctx, const zk_srs* srs, int basis, const void* const* d_scalar_ptrs, size_t count, size_t n, void* h_out_affine) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, (d_scalar_ptrs && srs && h_out_affine) || !count, "null pointer");
    if (count == 0) return ZK_OK;
    ZK_REQUIRE(ctx, basis == 0 || basis == 1, "basis must be 0 (monomial) or 1 (Lagrange)");
    ZK_REQUIRE(ctx, n <= ((size_t)1 << srs->k), "polynomial longer than the SRS");
    for (size_t i = 0; i < count; ++i) ZK_REQUIRE(ctx, d_scalar_ptrs[i] || !n, "null column pointer");
    // no hints from the caller: the columns are judged on the device (4096 sampled cells each, one launch per batch) and those with
    // at most a quarter of field-sized cells take the per-window path
    std::vector<uint8_t> narrow(count);
    if (int rc = zk::sample_narrow_dev(ctx, d_scalar_ptrs, count, n, narrow.data())) return rc;
    return zk::commit_batch_staged(ctx, srs, basis, d_scalar_ptrs, count, n, h_out_affine, nullptr, nullptr, narrow.data());
}
// The same with a hint per column (nullable): 1 = small integers (selectors, bytes, counters, lookup
// multiplicities), 2 = long runs of equal values (running products) -- see zkmi355.h.
int zk_commit_batch_hint(zk_ctx* ctx, const zk_srs* srs, int basis, const void* const* d_scalar_ptrs, size_t count, size_t n, const uint8_t* narrow, void* h_out_affine) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    return zk::commit_batch_staged(ctx, srs, basis, d_scalar_ptrs, count, n, h_out_affine, nullptr, nullptr, narrow);
}
// Same, for columns that still live in host memory: column i + 1 is uploaded on the copy stream
// while the MSM of column i runs.  d_cols[i] (n x 32 B each) receive the uploaded columns.
int zk_commit_batch_h2d(zk_ctx* ctx, const zk_srs* srs, int basis, const void* const* h_cols, void* const* d_cols, size_t count, size_t n, void* h_out_affine) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, (h_cols && d_cols) || !count, "null pointer");
    for (size_t i = 0; i < count; ++i) ZK_REQUIRE(ctx, h_cols[i] && d_cols[i], "null column pointer");
    struct Stage { zk_ctx* ctx; const void* const* h; void* const* d; size_t bytes; } st{ctx, h_cols, d_cols, n * sizeof(Fr)};
    int rc = zk::copy_stream_open(ctx);
    if (rc) return rc;
    auto fn = [](void* user, size_t it) -> int {
        Stage* s = (Stage*)user;
        ZK_HIP(s->ctx, hipMemcpyAsync(s->d[it], s->h[it], s->bytes, hipMemcpyHostToDevice, s->ctx->stream_copy));
        return zk::copy_stream_fence(s->ctx);
    };
    std::vector<uint8_t> narrow(count);
    zk::sample_narrow(h_cols, count, n, narrow.data());
    return zk::commit_batch_staged(ctx, srs, basis, (const void* const*)d_cols, count, n, h_out_affine, fn, &st, narrow.data());
}
int zk_msm_g1_host(zk_ctx* ctx, const void* h_scalars, const void* h_bases, size_t n, void* h_out_affine) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, h_out_affine && ((h_scalars && h_bases) || n == 0), "null pointer");
    if (n == 0) { memset(h_out_affine, 0, sizeof(G1Affine)); return ZK_OK; }
    char* d = (char*)ctx->get_scratch(SC_TMP2, n * (sizeof(Fr) + sizeof(G1Affine)));
    if (!d) return ZK_ERR_OOM;
    G1Affine* db = (G1Affine*)d;
    Fr* ds = (Fr*)(d + n * sizeof(G1Affine));
    ZK_HIP(ctx, hipMemcpyAsync(db, h_bases, n * sizeof(G1Affine), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipMemcpyAsync(ds, h_scalars, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    return msm_run(ctx, ds, db, n, (G1Affine*)h_out_affine);
}

}  // extern "C"

namespace zk {
// Which columns fill only a few Pippenger windows?  Witness columns are mostly small integers
// (bytes, flags, counters, selectors) and those take the per-window MSM path, which never touches
// the bucket sets of empty windows; only a sample is inspected (the choice affects speed, never the
// result).  Columns are in Montgomery form: a sampled value is converted back before it is judged.
// A column counts as small-valued when at most a quarter of the sampled cells are >= 2^64: the per-window path carries a
// minority of field-sized cells (RLC accumulators, hash outputs among bytes; SURVEY 8d's "60 % zero / 30 % < 2^16 / 10 % uniform")
// as a thin layer over all its windows and still beats the merged path -- measured per 2^20 column in a batch (tools/msm_dist.py):
// 10 % large cells 0.40 against 0.53 ms, 1 % 0.32 / 0.49, a third 0.70 / 0.69 (the break-even).
static constexpr size_t NARROW_MAX_LARGE_NUM = 1, NARROW_MAX_LARGE_DEN = 4;
void sample_narrow(const void* const* h_cols, size_t count, size_t n, uint8_t* narrow) {
    // 1024 samples per column, each one Montgomery product on the host (24 ns): 25 ms IN FRONT of the first upload of a
    // 1000-column advice phase on one thread -- on up to eight.
    const size_t samples = n < 1024 ? n : 1024, step = n / (samples ? samples : 1);
    auto judge = [&](size_t c0, size_t c1) {
        for (size_t c = c0; c < c1; ++c) {
            const host::F4* col = (const host::F4*)h_cols[c];
            size_t large = 0;
            const size_t allowed = samples * NARROW_MAX_LARGE_NUM / NARROW_MAX_LARGE_DEN;
            for (size_t i = 0; col && large <= allowed && i < samples; ++i) {
                const host::F4 v = host::fr_canon(col[i * step + (i * 7 + c) % (step ? step : 1)]);
                large += (v.l[1] | v.l[2] | v.l[3]) != 0;
            }
            narrow[c] = col && large <= allowed ? 1 : 0;
        }
    };
    const size_t threads = count >= 64 ? std::min<size_t>(8, std::max<size_t>(1, std::thread::hardware_concurrency())) : 1;
    if (threads <= 1) { judge(0, count); return; }
    std::vector<std::thread> pool;
    const size_t per = (count + threads - 1) / threads;
    size_t started = 0;                      // columns [0, started) have a thread
    try {
        for (size_t t = 0; t < threads && t * per < count; ++t) { pool.emplace_back(judge, t * per, std::min(count, (t + 1) * per)); started = std::min(count, (t + 1) * per); }
    } catch (...) {}                         // no thread to be had (a process at its limit): the rest is judged here; nothing unwinds across the C ABI
    if (started < count) judge(started, count);
    for (std::thread& th : pool) th.join();
}
// The same judgement for columns that are already on the device: one workgroup per column converts 4096 evenly spread cells
// and counts the ones >= 2^64; one launch and one small download per batch.
__global__ void __launch_bounds__(256) k_sample_large(const Fr* const* __restrict__ cols, uint64_t n, uint32_t* __restrict__ large) {
    const Fr* col = cols[blockIdx.x];
    const uint64_t samples = n < 4096 ? n : 4096, step = n / (samples ? samples : 1);
    uint32_t cnt = 0;
    for (uint64_t i = threadIdx.x; i < samples; i += 256) {
        const Fr v = from_mont(ldg(col + i * step + (i * 7 + blockIdx.x) % (step ? step : 1)));
        cnt += (v.l[2] | v.l[3] | v.l[4] | v.l[5] | v.l[6] | v.l[7]) != 0;
    }
    __shared__ uint32_t tot;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    atomicAdd(&tot, cnt);
    __syncthreads();
    if (threadIdx.x == 0) large[blockIdx.x] = tot;
}
int sample_narrow_dev(zk_ctx* ctx, const void* const* d_cols, size_t count, size_t n, uint8_t* narrow) {
    if (!count) return ZK_OK;
    char* scratch = (char*)ctx->get_scratch(SC_TMP, count * 12);
    if (!scratch) return ctx->fail(ZK_ERR_OOM, "column classification: scratch allocation failed");
    ZK_HIP(ctx, hipMemcpyAsync(scratch, d_cols, count * 8, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* d_large = (uint32_t*)(scratch + count * 8);
    hipLaunchKernelGGL(k_sample_large, dim3((unsigned)count), dim3(256), 0, ctx->stream, (const Fr* const*)scratch, (uint64_t)n, d_large);
    ZK_CHECK_LAUNCH(ctx);
    std::vector<uint32_t> large(count);
    ZK_HIP(ctx, hipMemcpyAsync(large.data(), d_large, count * 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t samples = n < 4096 ? n : 4096;
    for (size_t c = 0; c < count; ++c) narrow[c] = large[c] <= samples * NARROW_MAX_LARGE_NUM / NARROW_MAX_LARGE_DEN ? 1 : 0;
    return ZK_OK;
}
int commit_batch_staged(zk_ctx* ctx, const zk_srs* srs, int basis, const void* const* d_scalar_ptrs, size_t count, size_t n, void* h_out_affine, MsmStageFn stage, void* stage_user, const uint8_t* narrow) {
    if (count == 0) return ZK_OK;            // an empty batch (a circuit without permutation columns or lookups) commits nothing
    ZK_REQUIRE(ctx, srs && h_out_affine && d_scalar_ptrs, "null pointer");
    ZK_REQUIRE(ctx, basis == 0 || basis == 1, "basis must be 0 (monomial) or 1 (Lagrange)");      // index 2 (the prefix basis) is internal: msm_diff_try only
    ZK_REQUIRE(ctx, n <= ((size_t)1 << srs->k), "polynomial longer than the SRS");
    for (size_t i = 0; i < count; ++i) ZK_REQUIRE(ctx, d_scalar_ptrs[i] || !n, "null column pointer");
    const G1Affine* b = basis ? srs->g_lagrange : srs->g;
    ZK_REQUIRE(ctx, b, "SRS has no Lagrange basis");
    return msm_batch_srs(ctx, srs, basis, (const Fr* const*)d_scalar_ptrs, count, n, (G1Affine*)h_out_affine, stage, stage_user, narrow);
}
int copy_stream_open(zk_ctx* ctx) {
    if (!ctx->stream_copy) ZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_copy, hipStreamNonBlocking));
    if (!ctx->ev_copy) ZK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_copy, hipEventDisableTiming));      // the stream may predate it (zk_ctx_debug_delay)
    ZK_HIP(ctx, hipEventRecord(ctx->ev_copy, ctx->stream));
    ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream_copy, ctx->ev_copy, 0));
    return ZK_OK;
}
int copy_stream_fence(zk_ctx* ctx) {
    ZK_HIP(ctx, hipEventRecord(ctx->ev_copy, ctx->stream_copy));
    ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_copy, 0));
    return ZK_OK;
}
}  // namespace zk
