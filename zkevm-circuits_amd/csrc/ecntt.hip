// Lagrange basis of a KZG SRS from its monomial basis: halo2_proofs::poly::kzg::commitment
// `g_to_lagrange` (ParamsKZG::setup / from_parts / downsize; reference call sites
// prover/src/common/prover.rs:40-60 `params.downsize(k)`, prover/src/utils.rs:77 read_custom).
//
//      g_lagrange[i] = (1/n) * sum_j omega^(-i j) * g[j]        (an inverse FFT over G1)
//
// so that commit_lagrange(evaluations) = commit(coefficients).  Radix-2 DIT over XYZZ points in
// HBM (144 B each, ec29.hip.hpp arithmetic): log2(n) stages of n/2 butterflies, each one scalar
// multiplication by a 254-bit twiddle (double-and-add, ~4000 Montgomery products) and two point
// additions.  A one-off per SRS size: 2^20 points take about half a second; the reference's CPU
// version is the slow part of `downsize`.
#include "ctx.hpp"
#include "ec29.hip.hpp"

namespace zk {

__global__ void k_powers(Fr base, Fr mul, Fr* out, uint32_t count, int rprime);   // ntt.hip

__device__ __forceinline__ uint32_t bitrev32(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// affine, R = 2^256 form -> XYZZ on 29-bit limbs in R' = 2^261 form, stored bit-reversed (DIT input order)
__global__ void __launch_bounds__(256) k_ecntt_load(const G1Affine* __restrict__ g, G1Xyzz29* __restrict__ pts, uint64_t n, int log_n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine p = ldg(g + i);
    G1Xyzz29 q = identity29();
    if (!p.is_identity()) {
#pragma unroll
        for (int t = 0; t < 5; ++t) { p.x = dbl(p.x); p.y = dbl(p.y); }     // x 32: R -> R'
        q = G1Xyzz29{unpack29<Fq29P>(p.x), unpack29<Fq29P>(p.y), one29(), one29()};
    }
    stg29(pts + bitrev32((uint32_t)i, log_n), q);
}

// k * P, k a canonical 256-bit integer, MSB-first double-and-add
__device__ inline G1Xyzz29 mul_scalar29(const G1Xyzz29& p, const Fr& k) {
    int top = -1;
#pragma unroll 1
    for (int w = 7; w >= 0; --w) if (k.l[w]) { top = 32 * w + 31 - __clz(k.l[w]); break; }
    G1Xyzz29 acc = identity29();
#pragma unroll 1
    for (int bit = top; bit >= 0; --bit) {
        acc = dbl29pt(acc);
        if ((k.l[bit >> 5] >> (bit & 31)) & 1) acc = add29pt(acc, p);
    }
    return acc;
}
// -P for a point that keeps the stored invariant (0 < y < 8p: no 2-torsion on G1)
__device__ __forceinline__ G1Xyzz29 neg29pt(const G1Xyzz29& p) {
    if (is_identity29(p)) return p;
    G1Xyzz29 r = p;
    r.y = sub_n<8>(zero29(), p.y);
    return r;
}

// one DIT stage: (u, v) -> (u + w v, u - w v), w = tw[j << (log_n - 1 - s)] (Montgomery Fr)
__global__ void __launch_bounds__(256) k_ecntt_stage(G1Xyzz29* __restrict__ pts, const Fr* __restrict__ tw, int log_n, int s) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (1ull << (log_n - 1))) return;
    const uint64_t half = 1ull << s, j = b & (half - 1);
    const uint64_t i0 = ((b >> s) << (s + 1)) | j, i1 = i0 + half;
    const G1Xyzz29 u = ldg29(pts + i0);
    G1Xyzz29 v = ldg29(pts + i1);
    if (j) v = mul_scalar29(v, from_mont(ldg(tw + (j << (log_n - 1 - s)))));
    stg29(pts + i0, add29pt(u, v));
    stg29(pts + i1, add29pt(u, neg29pt(v)));
}

// out[i] = affine, R form, of scale * P[i]
__global__ void __launch_bounds__(256) k_ecntt_finish(const G1Xyzz29* __restrict__ pts, Fr scale_canon, G1Affine* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Xyzz29 p = mul_scalar29(ldg29(pts + i), scale_canon);
    if (is_identity29(p)) { stg(out + i, G1Affine{Fq::zero(), Fq::zero()}); return; }
    const Fq29 t = inv29(mul29(p.zz, p.zzz));
    const Fq29 izz = mul29(t, p.zzz), izzz = mul29(t, p.zz);
    const Fq29 c = unpack29<Fq29P>(Fq::one());            // the integer 2^256 mod p: R' -> R
    stg(out + i, G1Affine{pack29_lt2p(mul29(mul29(p.x, izz), c)), pack29_lt2p(mul29(mul29(p.y, izzz), c))});
}

// d_out[0 .. 2^k) = inverse group FFT of d_g[0 .. 2^k)  (d_out may not alias d_g)
int g_to_lagrange(zk_ctx* ctx, const G1Affine* d_g, uint32_t k, G1Affine* d_out) {
    const uint64_t n = 1ull << k;
    if (k == 0) {
        ZK_HIP(ctx, hipMemcpyAsync(d_out, d_g, sizeof(G1Affine), hipMemcpyDeviceToDevice, ctx->stream));
        return ZK_OK;
    }
    G1Xyzz29* pts = nullptr;
    Fr* tw = nullptr;
    if (hipMalloc(&pts, sizeof(G1Xyzz29) * n) != hipSuccess || hipMalloc(&tw, sizeof(Fr) * (n / 2)) != hipSuccess) {
        (void)hipGetLastError();
        if (pts) (void)hipFree(pts);
        return ctx->fail(ZK_ERR_OOM, "g_to_lagrange: allocation of %zu bytes failed", (size_t)(sizeof(G1Xyzz29) * n));
    }
    const Fr omega_inv = fr_inv_host(fr_root_of_unity(k));
    const Fr n_inv_canon = from_mont(fr_inv_host(fr_from_u64(n)));
    const dim3 t(256);
    hipLaunchKernelGGL(k_powers, dim3((unsigned)((n / 2 + 255) / 256)), t, 0, ctx->stream, omega_inv, Fr::one(), tw, (uint32_t)(n / 2), 0);
    hipLaunchKernelGGL(k_ecntt_load, dim3((unsigned)((n + 255) / 256)), t, 0, ctx->stream, d_g, pts, n, (int)k);
    for (uint32_t s = 0; s < k; ++s)
        hipLaunchKernelGGL(k_ecntt_stage, dim3((unsigned)((n / 2 + 255) / 256)), t, 0, ctx->stream, pts, (const Fr*)tw, (int)k, (int)s);
    hipLaunchKernelGGL(k_ecntt_finish, dim3((unsigned)((n + 255) / 256)), t, 0, ctx->stream, (const G1Xyzz29*)pts, n_inv_canon, d_out, n);
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(pts);
    (void)hipFree(tw);
    if (e != hipSuccess || e2 != hipSuccess) return ctx->fail(ZK_ERR_HIP, "g_to_lagrange failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return ZK_OK;
}

}  // namespace zk
