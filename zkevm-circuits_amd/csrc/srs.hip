// KZG structured reference string on device: halo2_proofs::poly::kzg::commitment::ParamsKZG
// (external crate).  Reference call sites:
//   ParamsKZG::setup / unsafe_setup_with_s  [REF circuit-benchmarks/src/super_circuit.rs:104]
//                                           [REF zkevm-circuits/src/super_circuit/test.rs:74]
//   read_custom (host side, uploads here)   [REF prover/src/utils.rs:77]
// g[i] = s^i * G,  g_lagrange[i] = L_i(s) * G with L_i(s) = omega^i (s^n - 1) / (n (s - omega^i)).
//
// Fixed-base scalar multiplication uses an 8-bit windowed table of G (32 x 256 affine points,
// 512 KiB, L2-resident): 32 mixed additions per output point, no doublings.
#include "ctx.hpp"
#include "host_fq.hpp"

namespace zk {

__global__ void k_powers(Fr base, Fr mul, Fr* out, uint32_t count, int rprime);   // ntt.hip

constexpr int FB_WINDOWS = 32;   // 256 bits / 8

// table[j*256 + d] = d * 2^(8j) * G  (affine; d = 0 -> identity).  base[j] = 2^(8j) * G affine.
__global__ void k_fb_table(const G1Affine* __restrict__ base, G1Affine* __restrict__ table) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= FB_WINDOWS * 256) return;
    const uint32_t j = t >> 8, d = t & 255;
    const G1Affine b = ldg(base + j);
    G1Xyzz acc = G1Xyzz::identity();
    for (int bit = 7; bit >= 0; --bit) {
        acc = dbl(acc);
        if ((d >> bit) & 1) acc = madd(acc, b);
    }
    stg(table + t, to_affine(acc));
}

// out[i] = scalars[i] * G   (scalars Montgomery-form Fr), affine out
__global__ void __launch_bounds__(256) k_fb_mul(const Fr* __restrict__ scalars, const G1Affine* __restrict__ table, G1Affine* __restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr s = from_mont(ldg(scalars + i));
    G1Xyzz acc = G1Xyzz::identity();
#pragma unroll 1
    for (int j = 0; j < FB_WINDOWS; ++j) {
        const uint32_t d = (s.l[j >> 2] >> ((j & 3) * 8)) & 255u;
        if (d) acc = madd(acc, ldg(table + j * 256 + d));
    }
    stg(out + i, to_affine(acc));
}

// lag[i] = omega^i * c / (n * (s - omega^i)) is assembled as:  den[i] = s - w[i]   (then batch
// inverted by the caller)  and  lag[i] = w[i] * c * deninv[i]
__global__ void k_lagrange_den(const Fr* __restrict__ w, Fr s, Fr* __restrict__ den, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stg(den + i, s - ldg(w + i));
}
__global__ void k_lagrange_fin(const Fr* __restrict__ w, const Fr* __restrict__ deninv, Fr c, Fr* __restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stg(out + i, ldg(w + i) * c * ldg(deninv + i));
}

// variable-base: out[i] = scalars[i] * bases[i]   (tests of the group law)
__global__ void __launch_bounds__(256) k_g1_mul(const G1Affine* __restrict__ bases, const Fr* __restrict__ scalars, G1Affine* __restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr s = from_mont(ldg(scalars + i));
    const G1Affine b = ldg(bases + i);
    G1Xyzz acc = G1Xyzz::identity();
#pragma unroll 1
    for (int bit = 255; bit >= 0; --bit) {
        acc = dbl(acc);
        if ((s.l[bit >> 5] >> (bit & 31)) & 1) acc = madd(acc, b);
    }
    stg(out + i, to_affine(acc));
}
__global__ void __launch_bounds__(256) k_g1_add(const G1Affine* __restrict__ a, const G1Affine* __restrict__ b, G1Affine* __restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    stg(out + i, to_affine(madd(from_affine(ldg(a + i)), ldg(b + i))));
}

static int fixed_base_table(zk_ctx* ctx, G1Affine** table_out) {
    // 2^(8j) * G on the host (256 dependent doublings, then one affine conversion each)
    std::vector<G1Affine> base(FB_WINDOWS);
    host::PXyzz cur;
    memset(&cur, 0, sizeof cur);
    const host::F4 one = host::fone<host::FqC>();
    host::F4 gx = one, gy = host::fadd<host::FqC>(one, one);   // G = (1, 2) in Montgomery form
    cur.x = gx; cur.y = gy; cur.zz = one; cur.zzz = one;
    for (int j = 0; j < FB_WINDOWS; ++j) {
        host::pto_affine(cur, &base[j]);
        for (int i = 0; i < 8; ++i) cur = host::pdbl(cur);
    }
    G1Affine* d = (G1Affine*)ctx->get_scratch(SC_MSM_MISC, sizeof(G1Affine) * (FB_WINDOWS * 256 + FB_WINDOWS));
    if (!d) return ZK_ERR_OOM;
    G1Affine* d_base = d + FB_WINDOWS * 256;
    ZK_HIP(ctx, hipMemcpyAsync(d_base, base.data(), sizeof(G1Affine) * FB_WINDOWS, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `base` is a stack-owned staging buffer
    hipLaunchKernelGGL(k_fb_table, dim3(FB_WINDOWS), dim3(256), 0, ctx->stream, (const G1Affine*)d_base, d);
    ZK_CHECK_LAUNCH(ctx);
    *table_out = d;
    return ZK_OK;
}

}  // namespace zk

using namespace zk;

extern "C" {

int zk_srs_create(zk_ctx* ctx, uint32_t k, const void* h_g, const void* h_g_lagrange, zk_srs** out) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, out && h_g, "null pointer");
    ZK_REQUIRE(ctx, k <= 28, "k exceeds the two-adicity of Fr (28)");
    const size_t n = (size_t)1 << k;
    zk_srs* s = new zk_srs();
    s->k = k;
    if (hipMalloc(&s->g, sizeof(G1Affine) * n) != hipSuccess) { delete s; return ctx->fail(ZK_ERR_OOM, "SRS allocation failed"); }
    ZK_HIP(ctx, hipMemcpyAsync(s->g, h_g, sizeof(G1Affine) * n, hipMemcpyHostToDevice, ctx->stream));
    if (hipMalloc(&s->g_lagrange, sizeof(G1Affine) * n) != hipSuccess) { (void)hipFree(s->g); delete s; return ctx->fail(ZK_ERR_OOM, "SRS allocation failed"); }
    if (h_g_lagrange) {
        ZK_HIP(ctx, hipMemcpyAsync(s->g_lagrange, h_g_lagrange, sizeof(G1Affine) * n, hipMemcpyHostToDevice, ctx->stream));
    } else {
        // ParamsKZG::from_parts with g_lagrange = None: derive it (inverse FFT over G1, ecntt.hip)
        int rc = g_to_lagrange(ctx, s->g, k, s->g_lagrange);
        if (rc) { zk_srs_destroy(ctx, s); return rc; }
    }
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = s;
    return ZK_OK;
}

// ParamsKZG::downsize: the first 2^new_k monomial-basis points with the Lagrange basis of the smaller domain
int zk_srs_downsize(zk_ctx* ctx, const zk_srs* srs, uint32_t new_k, zk_srs** out) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, srs && out, "null pointer");
    ZK_REQUIRE(ctx, new_k <= srs->k, "downsize can only shrink the SRS");
    const size_t n = (size_t)1 << new_k;
    zk_srs* s = new zk_srs();
    s->k = new_k;
    if (hipMalloc(&s->g, sizeof(G1Affine) * n) != hipSuccess || hipMalloc(&s->g_lagrange, sizeof(G1Affine) * n) != hipSuccess) {
        (void)hipGetLastError();
        zk_srs_destroy(ctx, s);
        return ctx->fail(ZK_ERR_OOM, "SRS allocation failed");
    }
    ZK_HIP(ctx, hipMemcpyAsync(s->g, srs->g, sizeof(G1Affine) * n, hipMemcpyDeviceToDevice, ctx->stream));
    int rc = ZK_OK;
    if (new_k == srs->k && srs->g_lagrange) ZK_HIP(ctx, hipMemcpyAsync(s->g_lagrange, srs->g_lagrange, sizeof(G1Affine) * n, hipMemcpyDeviceToDevice, ctx->stream));
    else rc = g_to_lagrange(ctx, s->g, new_k, s->g_lagrange);
    if (rc) { zk_srs_destroy(ctx, s); return rc; }
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = s;
    return ZK_OK;
}

int zk_srs_setup_with_s(zk_ctx* ctx, uint32_t k, const void* h_s, zk_srs** out) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, out && h_s, "null pointer");
    ZK_REQUIRE(ctx, k <= 28, "k exceeds the two-adicity of Fr (28)");
    const uint64_t n = 1ull << k;
    const Fr s = *(const Fr*)h_s;
    zk_srs* r = new zk_srs();
    r->k = k;
    if (hipMalloc(&r->g, sizeof(G1Affine) * n) != hipSuccess || hipMalloc(&r->g_lagrange, sizeof(G1Affine) * n) != hipSuccess) {
        if (r->g) (void)hipFree(r->g);
        delete r;
        return ctx->fail(ZK_ERR_OOM, "SRS allocation failed");
    }
    G1Affine* table = nullptr;
    int rc = fixed_base_table(ctx, &table);
    if (rc) { zk_srs_destroy(ctx, r); return rc; }
    Fr* sc = (Fr*)ctx->get_scratch(SC_TMP2, sizeof(Fr) * n * 2);
    if (!sc) { zk_srs_destroy(ctx, r); return ZK_ERR_OOM; }
    Fr* w = sc + n;
    const dim3 g((unsigned)((n + 255) / 256)), t(256);
    // g[i] = s^i * G
    hipLaunchKernelGGL(k_powers, g, t, 0, ctx->stream, s, Fr::one(), sc, (uint32_t)n, 0);
    hipLaunchKernelGGL(k_fb_mul, g, t, 0, ctx->stream, (const Fr*)sc, (const G1Affine*)table, r->g, n);
    // g_lagrange[i] = omega^i (s^n - 1) / (n (s - omega^i)) * G
    const Fr omega = fr_root_of_unity(k);
    Fr sn = s;
    for (uint32_t i = 0; i < k; ++i) sn = sqr(sn);
    const Fr c = (sn - Fr::one()) * fr_inv_host(fr_from_u64(n));
    hipLaunchKernelGGL(k_powers, g, t, 0, ctx->stream, omega, Fr::one(), w, (uint32_t)n, 0);
    hipLaunchKernelGGL(k_lagrange_den, g, t, 0, ctx->stream, (const Fr*)w, s, sc, n);
    ZK_CHECK_LAUNCH(ctx);
    rc = zk_fr_batch_invert(ctx, sc, n);
    if (rc) { zk_srs_destroy(ctx, r); return rc; }
    hipLaunchKernelGGL(k_lagrange_fin, g, t, 0, ctx->stream, (const Fr*)w, (const Fr*)sc, c, sc, n);
    hipLaunchKernelGGL(k_fb_mul, g, t, 0, ctx->stream, (const Fr*)sc, (const G1Affine*)table, r->g_lagrange, n);
    ZK_CHECK_LAUNCH(ctx);
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = r;
    return ZK_OK;
}

void zk_srs_destroy(zk_ctx* ctx, zk_srs* srs) {
    (void)ctx;
    if (!srs) return;
    if (srs->g) (void)hipFree(srs->g);
    if (srs->g_lagrange) (void)hipFree(srs->g_lagrange);
    if (srs->g_rp) (void)hipFree(srs->g_rp);
    if (srs->g_lagrange_rp) (void)hipFree(srs->g_lagrange_rp);
    for (auto t : srs->tab) if (t) (void)hipFree(t);
    for (auto t : srs->tabn) if (t) (void)hipFree(t);
    for (auto t : srs->pfx) if (t) (void)hipFree(t);
    for (auto t : srs->pfx_negtot) if (t) (void)hipFree(t);
    for (auto t : srs->pfx_negtot_tab) if (t) (void)hipFree(t);
    delete srs;
}
uint32_t zk_srs_k(const zk_srs* srs) { return srs ? srs->k : 0; }
const void* zk_srs_g(const zk_srs* srs) { return srs ? srs->g : nullptr; }
const void* zk_srs_g_lagrange(const zk_srs* srs) { return srs ? srs->g_lagrange : nullptr; }

int zk_g1_affine_add_vec(zk_ctx* ctx, const void* d_a, const void* d_b, void* d_out, size_t n) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_a && d_b && d_out, "null pointer");
    if (!n) return ZK_OK;
    hipLaunchKernelGGL(k_g1_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const G1Affine*)d_a, (const G1Affine*)d_b, (G1Affine*)d_out, (uint64_t)n);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}
int zk_g1_mul_vec(zk_ctx* ctx, const void* d_bases, const void* d_scalars, void* d_out, size_t n) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_bases && d_scalars && d_out, "null pointer");
    if (!n) return ZK_OK;
    hipLaunchKernelGGL(k_g1_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const G1Affine*)d_bases, (const Fr*)d_scalars, (G1Affine*)d_out, (uint64_t)n);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}

}  // extern "C"
