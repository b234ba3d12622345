// SRS files: halo2_proofs ParamsKZG::{read_custom, write_custom} (external crate).
// The reference loads `params{k}` through prover::utils::load_params, which checks the file length
// against the degree before parsing  [REF prover/src/utils.rs:39-84]  and defaults to
// SerdeFormat::RawBytesUnchecked  [REF prover/src/utils.rs:32].
//
//   file = u32 k (LE) | g[0..n) | g_lagrange[0..n) | g2 | s_g2
//   G1: 64 B (RawBytes / RawBytesUnchecked: x | y, Montgomery limbs = the in-memory image, goes to
//       the device as is) or 32 B (Processed: x canonical LE, bit 254 = parity of y, bit 255 = identity)
//   G2: twice the G1 size.  The prover never touches G2; the two blobs are handed through.
//
// Decompression (one (p+1)/4 power per point), on-curve checks and compression run on the device:
// 2^21 square roots are a few milliseconds there and half a minute on one host core.
// zk_g2_setup (host only) gives unsafe_setup_with_s its G2 half: the generator and s * generator.
#include <vector>

#include "ctx.hpp"
#include "host_fq.hpp"

namespace zk {

// a^((p+1)/4): the square root candidate in Fq (p = 3 mod 4)
__device__ __forceinline__ Fq fq_sqrt_candidate(const Fq& a) {
    // (p + 1) / 4, little-endian u32 limbs
    const uint32_t e[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
    Fq r = Fq::one();
#pragma unroll 1
    for (int bit = 251; bit >= 0; --bit) {
        r = sqr(r);
        if ((e[bit >> 5] >> (bit & 31)) & 1) r = r * a;
    }
    return r;
}
__device__ __forceinline__ bool fq_canonical(const Fq& a) {   // raw limbs below the modulus
    uint32_t m[8];
    load_mod<FqP>(m);
    for (int i = 7; i >= 0; --i) {
        if (a.l[i] < m[i]) return true;
        if (a.l[i] > m[i]) return false;
    }
    return false;
}
__device__ __forceinline__ bool g1_on_curve(const G1Affine& p) {
    Fq b3 = Fq::one();
    b3 = b3 + b3 + b3;                                        // 3 in Montgomery form
    return sqr(p.y) == sqr(p.x) * p.x + b3;
}

// bad[0] counts rejected encodings (x >= p, no square root, an identity flag on a non-zero image); halo2curves' flag bits:
// bit 254 = parity of y, bit 255 = identity (host_util.hpp g1_compress has the provenance)
__global__ void __launch_bounds__(256) k_g1_decompress(const uint8_t* __restrict__ in, G1Affine* __restrict__ out, uint64_t n, uint32_t* __restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* q = reinterpret_cast<const uint4*>(in + i * 32);
    const uint4 lo = q[0], hi = q[1];
    Fq x{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    const uint32_t is_inf = x.l[7] >> 31, ysign = (x.l[7] >> 30) & 1u;
    x.l[7] &= 0x3FFFFFFFu;
    G1Affine p{Fq::zero(), Fq::zero()};
    if (is_inf) { if (ysign || !x.is_zero()) atomicAdd(bad, 1u); stg(out + i, p); return; }
    if (!fq_canonical(x)) { atomicAdd(bad, 1u); stg(out + i, p); return; }
    x = to_mont(x);
    Fq b3 = Fq::one();
    b3 = b3 + b3 + b3;
    const Fq rhs = sqr(x) * x + b3;
    Fq y = fq_sqrt_candidate(rhs);
    if (sqr(y) != rhs) { atomicAdd(bad, 1u); stg(out + i, p); return; }
    if ((from_mont(y).l[0] & 1u) != ysign) y = neg(y);
    p.x = x;
    p.y = y;
    stg(out + i, p);
}
__global__ void __launch_bounds__(256) k_g1_compress(const G1Affine* __restrict__ in, uint8_t* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine p = ldg(in + i);
    Fq x = Fq::zero();
    if (!p.is_identity()) {
        x = from_mont(p.x);
        x.l[7] |= (from_mont(p.y).l[0] & 1u) << 30;
    } else {
        x.l[7] = 0x80000000u;
    }
    uint4* q = reinterpret_cast<uint4*>(out + i * 32);
    q[0] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
    q[1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
}
// SerdeFormat::RawBytes reads check what RawBytesUnchecked skips: limbs below p, point on the curve
__global__ void __launch_bounds__(256) k_g1_check(const G1Affine* __restrict__ pts, uint64_t n, uint32_t* __restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine p = ldg(pts + i);
    if (!fq_canonical(p.x) || !fq_canonical(p.y) || !(p.is_identity() || g1_on_curve(p))) atomicAdd(bad, 1u);
}

// ---- host: Fq2 = Fq[u] / (u^2 + 1) and the G2 group law (XYZZ formulas hold over any field, a = 0)
namespace host {
struct F2 { F4 c0, c1; };
typedef FqC C;
inline F2 f2_add(const F2& a, const F2& b) { return F2{fadd<C>(a.c0, b.c0), fadd<C>(a.c1, b.c1)}; }
inline F2 f2_sub(const F2& a, const F2& b) { return F2{fsub<C>(a.c0, b.c0), fsub<C>(a.c1, b.c1)}; }
inline F2 f2_mul(const F2& a, const F2& b) {
    const F4 t0 = fmul<C>(a.c0, b.c0), t1 = fmul<C>(a.c1, b.c1);
    const F4 s = fmul<C>(fadd<C>(a.c0, a.c1), fadd<C>(b.c0, b.c1));
    return F2{fsub<C>(t0, t1), fsub<C>(fsub<C>(s, t0), t1)};
}
inline bool f2_zero(const F2& a) { return fzero<C>(a.c0) && fzero<C>(a.c1); }
inline F2 f2_inv(const F2& a) {                        // conj(a) / (c0^2 + c1^2)
    const F4 t = finv<C>(fadd<C>(fmul<C>(a.c0, a.c0), fmul<C>(a.c1, a.c1)));
    F4 z; memset(&z, 0, sizeof z);
    return F2{fmul<C>(a.c0, t), fmul<C>(fsub<C>(z, a.c1), t)};
}
struct Q2 { F2 x, y, zz, zzz; };
inline bool q2_id(const Q2& p) { return f2_zero(p.zz); }
inline Q2 q2_dbl(const Q2& p) {
    if (q2_id(p)) return p;
    const F2 u = f2_add(p.y, p.y), v = f2_mul(u, u), w = f2_mul(u, v), s = f2_mul(p.x, v);
    const F2 x2 = f2_mul(p.x, p.x), m = f2_add(f2_add(x2, x2), x2);
    Q2 r;
    r.x = f2_sub(f2_mul(m, m), f2_add(s, s));
    r.y = f2_sub(f2_mul(m, f2_sub(s, r.x)), f2_mul(w, p.y));
    r.zz = f2_mul(v, p.zz);
    r.zzz = f2_mul(w, p.zzz);
    return r;
}
inline Q2 q2_add(const Q2& p, const Q2& q) {
    if (q2_id(q)) return p;
    if (q2_id(p)) return q;
    const F2 u1 = f2_mul(p.x, q.zz), u2 = f2_mul(q.x, p.zz), s1 = f2_mul(p.y, q.zzz), s2 = f2_mul(q.y, p.zzz);
    const F2 P = f2_sub(u2, u1), R = f2_sub(s2, s1);
    if (f2_zero(P)) {
        if (f2_zero(R)) return q2_dbl(p);
        Q2 id; memset(&id, 0, sizeof id); return id;
    }
    const F2 pp = f2_mul(P, P), ppp = f2_mul(P, pp), qq = f2_mul(u1, pp);
    Q2 r;
    r.x = f2_sub(f2_sub(f2_mul(R, R), ppp), f2_add(qq, qq));
    r.y = f2_sub(f2_mul(R, f2_sub(qq, r.x)), f2_mul(s1, ppp));
    r.zz = f2_mul(f2_mul(p.zz, q.zz), pp);
    r.zzz = f2_mul(f2_mul(p.zzz, q.zzz), ppp);
    return r;
}
inline F4 fq_from_words(const uint64_t (&w)[4]) {         // canonical -> Montgomery: times R^2, reduced once
    static const uint64_t R2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL};
    F4 a, r2; memcpy(a.l, w, 32); memcpy(r2.l, R2, 32);
    return fmul<C>(a, r2);
}
}  // namespace host
}  // namespace zk

using namespace zk;

static size_t g1_len(int format) { return format == ZK_SERDE_PROCESSED ? 32 : 64; }

extern "C" {

size_t zk_params_file_len(uint32_t k, int format) {
    if (k > 28 || format < 0 || format > 2) return 0;
    return 4 + 2 * ((size_t)1 << k) * g1_len(format) + 2 * 2 * g1_len(format);
}

int zk_params_read(zk_ctx* ctx, const void* h_file, size_t len, int format, zk_srs** out, void* h_g2, void* h_s_g2) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, h_file && out, "null pointer");
    ZK_REQUIRE(ctx, format >= 0 && format <= 2, "format: 0 Processed, 1 RawBytes, 2 RawBytesUnchecked");
    ZK_REQUIRE(ctx, len >= 4, "params file shorter than its header");
    const uint8_t* f = (const uint8_t*)h_file;
    const uint32_t k = (uint32_t)f[0] | (uint32_t)f[1] << 8 | (uint32_t)f[2] << 16 | (uint32_t)f[3] << 24;
    ZK_REQUIRE(ctx, k <= 28, "params file: k exceeds the two-adicity of Fr (28)");
    // the reference refuses a file whose length does not match its degree before parsing it
    if (len != zk_params_file_len(k, format))
        return ctx->fail(ZK_ERR_INVALID_ARG, "invalid params file len %zu for degree %u (expected %zu): check the degree / serde format", len, k, zk_params_file_len(k, format));
    const size_t n = (size_t)1 << k, gl = g1_len(format);
    const uint8_t *fg = f + 4, *fl = fg + n * gl, *f2 = fl + n * gl;
    zk_srs* s = new zk_srs();
    s->k = k;
    uint32_t* d_bad = nullptr;
    uint8_t* d_in = nullptr;
    auto fail = [&](int code) { (void)hipFree(d_bad); (void)hipFree(d_in); zk_srs_destroy(ctx, s); return code; };
    if (hipMalloc(&s->g, sizeof(G1Affine) * n) != hipSuccess || hipMalloc(&s->g_lagrange, sizeof(G1Affine) * n) != hipSuccess || hipMalloc(&d_bad, 4) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx->fail(ZK_ERR_OOM, "SRS allocation failed"));
    }
    hipError_t e = hipMemsetAsync(d_bad, 0, 4, ctx->stream);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (format == ZK_SERDE_PROCESSED) {
        if (e == hipSuccess && hipMalloc(&d_in, 2 * n * 32) != hipSuccess) { (void)hipGetLastError(); return fail(ctx->fail(ZK_ERR_OOM, "SRS staging allocation failed")); }
        if (e == hipSuccess) e = hipMemcpyAsync(d_in, fg, 2 * n * 32, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_g1_decompress, grid, block, 0, ctx->stream, (const uint8_t*)d_in, s->g, (uint64_t)n, d_bad);
            hipLaunchKernelGGL(k_g1_decompress, grid, block, 0, ctx->stream, (const uint8_t*)d_in + n * 32, s->g_lagrange, (uint64_t)n, d_bad);
            e = hipGetLastError();
        }
    } else {
        if (e == hipSuccess) e = hipMemcpyAsync(s->g, fg, n * 64, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(s->g_lagrange, fl, n * 64, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && format == ZK_SERDE_RAW) {
            hipLaunchKernelGGL(k_g1_check, grid, block, 0, ctx->stream, (const G1Affine*)s->g, (uint64_t)n, d_bad);
            hipLaunchKernelGGL(k_g1_check, grid, block, 0, ctx->stream, (const G1Affine*)s->g_lagrange, (uint64_t)n, d_bad);
            e = hipGetLastError();
        }
    }
    uint32_t bad = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx->fail(ZK_ERR_HIP, "params read: %s", hipGetErrorString(e)));
    if (bad) return fail(ctx->fail(ZK_ERR_INVALID_ARG, "params file: %u G1 encodings are not points of the curve", bad));
    if (h_g2) memcpy(h_g2, f2, 2 * gl);
    if (h_s_g2) memcpy(h_s_g2, f2 + 2 * gl, 2 * gl);
    (void)hipFree(d_bad);
    (void)hipFree(d_in);
    *out = s;
    return ZK_OK;
}

int zk_params_write(zk_ctx* ctx, const zk_srs* srs, const void* h_g2, const void* h_s_g2, int format, void* h_out, size_t cap, size_t* len) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, srs && h_g2 && h_s_g2 && len, "null pointer");
    ZK_REQUIRE(ctx, format >= 0 && format <= 2, "format: 0 Processed, 1 RawBytes, 2 RawBytesUnchecked");
    ZK_REQUIRE(ctx, srs->g_lagrange, "SRS without a Lagrange basis");
    const size_t need = zk_params_file_len(srs->k, format);
    *len = need;
    if (!h_out) return ZK_OK;                                  // size query
    if (cap < need) return ctx->fail(ZK_ERR_INVALID_ARG, "params buffer of %zu bytes, %zu needed", cap, need);
    const size_t n = (size_t)1 << srs->k, gl = g1_len(format);
    uint8_t* o = (uint8_t*)h_out;
    o[0] = (uint8_t)srs->k; o[1] = o[2] = o[3] = 0;
    if (format == ZK_SERDE_PROCESSED) {
        uint8_t* d_out = nullptr;
        if (hipMalloc(&d_out, 2 * n * 32) != hipSuccess) { (void)hipGetLastError(); return ctx->fail(ZK_ERR_OOM, "SRS staging allocation failed"); }
        const dim3 grid((unsigned)((n + 255) / 256)), block(256);
        hipLaunchKernelGGL(k_g1_compress, grid, block, 0, ctx->stream, (const G1Affine*)srs->g, d_out, (uint64_t)n);
        hipLaunchKernelGGL(k_g1_compress, grid, block, 0, ctx->stream, (const G1Affine*)srs->g_lagrange, d_out + n * 32, (uint64_t)n);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(o + 4, d_out, 2 * n * 32, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_out);
        if (e != hipSuccess) return ctx->fail(ZK_ERR_HIP, "params write: %s", hipGetErrorString(e));
    } else {
        ZK_HIP(ctx, hipMemcpyAsync(o + 4, srs->g, n * 64, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(ctx, hipMemcpyAsync(o + 4 + n * 64, srs->g_lagrange, n * 64, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    memcpy(o + 4 + 2 * n * gl, h_g2, 2 * gl);
    memcpy(o + 4 + 2 * n * gl + 2 * gl, h_s_g2, 2 * gl);
    return ZK_OK;
}

// ParamsKZG::unsafe_setup_with_s, G2 half: the generator and s * generator as RawBytes
// (x.c0 | x.c1 | y.c0 | y.c1, Montgomery limbs).  Host only: one 254-bit double-and-add.
int zk_g2_setup(const void* h_s, void* h_g2, void* h_s_g2) {
    if (!h_s || !h_g2 || !h_s_g2) return ZK_ERR_INVALID_ARG;
    using namespace zk::host;
    // the alt_bn128 G2 generator (EIP-197; halo2curves bn256::G2::generator), canonical words LE
    static const uint64_t GX0[4] = {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL};
    static const uint64_t GX1[4] = {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL};
    static const uint64_t GY0[4] = {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL};
    static const uint64_t GY1[4] = {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL};
    Q2 g;
    g.x = F2{fq_from_words(GX0), fq_from_words(GX1)};
    g.y = F2{fq_from_words(GY0), fq_from_words(GY1)};
    F4 z; memset(&z, 0, sizeof z);
    g.zz = F2{fone<FqC>(), z};
    g.zzz = g.zz;
    memcpy(h_g2, &g.x, 64);
    memcpy((uint8_t*)h_g2 + 64, &g.y, 64);
    F4 s; memcpy(s.l, h_s, 32);
    F4 one; memset(&one, 0, sizeof one); one.l[0] = 1;
    s = fmul<FrC>(s, one);                                    // Montgomery -> canonical
    Q2 acc; memset(&acc, 0, sizeof acc);
    for (int bit = 255; bit >= 0; --bit) {
        acc = q2_dbl(acc);
        if ((s.l[bit >> 6] >> (bit & 63)) & 1) acc = q2_add(acc, g);
    }
    if (q2_id(acc)) { memset(h_s_g2, 0, 128); return ZK_OK; }
    const F2 t = f2_inv(f2_mul(acc.zz, acc.zzz));
    const F2 izz = f2_mul(t, acc.zzz), izzz = f2_mul(t, acc.zz);
    const F2 x = f2_mul(acc.x, izz), y = f2_mul(acc.y, izzz);
    memcpy(h_s_g2, &x, 64);
    memcpy((uint8_t*)h_s_g2 + 64, &y, 64);
    return ZK_OK;
}

}  // extern "C"
