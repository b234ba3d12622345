// Element-wise and scan kernels over Fr / Fq columns (HBM-streaming side of the prover):
//   field vector add/sub/mul            halo2curves Add/Sub/Mul           
//   batch inversion                     ff::BatchInvert                    (K12, K7, K8)
//   prefix product / prefix sum         permutation & lookup grand product (K7, K8)
//   eval_polynomial, kate_division      halo2_proofs::arithmetic           (K9, K10, K11)
// All kernels read/write 32-byte elements as two 16-byte transactions per lane, consecutive
// lanes on consecutive elements (fully coalesced), grid-stride where a fixed grid is needed.
#include "ctx.hpp"
#include "ff29.hip.hpp"

namespace zk {

int fr_scale_run(zk_ctx* ctx, Fr* d_a, const Fr& s, uint64_t n);   // ntt.hip
__global__ void k_powers(Fr base, Fr mul, Fr* out, uint32_t count, int rprime); // ntt.hip

template <class F, int OP>
__global__ void k_vec_op(const F* __restrict__ a, const F* __restrict__ b, F* __restrict__ o, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F x = ldg(a + i), y = ldg(b + i), r;
    if (OP == ZK_OP_ADD) r = x + y;
    else if (OP == ZK_OP_SUB) r = x - y;
    else r = x * y;
    stg(o + i, r);
}

template <class F>
static int vec_op_launch(zk_ctx* ctx, int op, const void* a, const void* b, void* o, uint64_t n) {
    dim3 g((unsigned)((n + 255) / 256)), t(256);
    const F* A = (const F*)a; const F* B = (const F*)b; F* O = (F*)o;
    switch (op) {
        case ZK_OP_ADD: hipLaunchKernelGGL((k_vec_op<F, ZK_OP_ADD>), g, t, 0, ctx->stream, A, B, O, n); break;
        case ZK_OP_SUB: hipLaunchKernelGGL((k_vec_op<F, ZK_OP_SUB>), g, t, 0, ctx->stream, A, B, O, n); break;
        case ZK_OP_MUL: hipLaunchKernelGGL((k_vec_op<F, ZK_OP_MUL>), g, t, 0, ctx->stream, A, B, O, n); break;
        default: return ctx->fail(ZK_ERR_INVALID_ARG, "unknown vector op %d", op);
    }
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}

// ---------------------------------------------------------------------------- batch inversion
// Lane t owns the strided set {t, t + NT, ...} of BI_K elements (coalesced), all held in registers: prefix products inside the
// lane, a product scan across the 64 lanes of the wave (shuffles), ONE Fermat inversion per wave, and back: every lane gets the
// inverse of its own product from the wave's, then walks its elements backwards.  Zeros are skipped (stay zero).
// Round 3: the first form gave every thread 32 elements and an inversion of its own -- 0.5 waves per SIMD at 2^20, two dependent
// global loads per element and step with nothing to hide them behind, an inversion per 32 elements: 290 us whatever the size
// (36 ms of the SuperCircuit-shape proof in 122 calls) -- most of it the inversion itself: a wave issues its ~95 000 instructions
// at the same pace whether one lane or all of them are live.  Now: 8 + 12 + 8 products on 29-bit limbs around ONE inversion per
// wave by binary extended Euclid (inv_xgcd, a quarter of the instructions), loads issued up front.
constexpr int BI_K = 8;
__device__ __forceinline__ Fr29 shfl_up29(const Fr29& v, int off) {
    Fr29 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = __shfl_up(v.l[k], off);
    return r;
}
__device__ __forceinline__ Fr29 shfl_down29(const Fr29& v, int off) {
    Fr29 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = __shfl_down(v.l[k], off);
    return r;
}
__device__ __forceinline__ Fr29 shfl29(const Fr29& v, int src) {
    Fr29 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = __shfl(v.l[k], src);
    return r;
}
// 1 / b for b in R' = 2^261 form, result in R' form: the residue as a plain integer X = B 2^261 has X^-1 = B^-1 2^-261 (binary
// extended Euclid, ff.hip.hpp), and one product by c783 = 2^783 brings it to B^-1 2^261.  A lone lane runs this while its wave
// waits: the Fermat ladder (381 products, ~95 000 instructions) took 250 us there, this takes a quarter of the instructions.
__device__ inline Fr29 inv29_rp(const Fr29& b, const Fr29& c783) {
    const Fr x = pack29_lt2p(b);
    Fr y;
    inv_xgcd<FrP>(y.l, x.l);
    return mul29(unpack29<Fr29P>(y), c783);
}
__global__ void __launch_bounds__(256) k_batch_invert(Fr* __restrict__ a, uint64_t n, uint64_t nt, Fr c783_plain) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;                       // nt is a multiple of 64: whole waves leave
    const uint32_t lane = threadIdx.x & 63u;
    auto rprime = [](Fr x) { for (int i = 0; i < 5; ++i) x = dbl(x); return x; };       // R -> R' = 2^261 form: x 32
    const Fr29 one_rp = unpack29<Fr29P>(rprime(Fr::one()));
    Fr x[BI_K];
#pragma unroll
    for (int k = 0; k < BI_K; ++k) {           // every load in flight before the first product
        const uint64_t i = (uint64_t)k * nt + t;
        x[k] = i < n ? ldg(a + i) : Fr::zero();
    }
    Fr29 v[BI_K], pre[BI_K];
    Fr29 acc = one_rp;
#pragma unroll
    for (int k = 0; k < BI_K; ++k) {
        pre[k] = acc;
        if (!x[k].is_zero()) { v[k] = unpack29<Fr29P>(rprime(x[k])); acc = mul29(acc, v[k]); }
    }
    // inclusive products over the lanes from both ends
    Fr29 up = acc, down = acc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Fr29 u = shfl_up29(up, off), d = shfl_down29(down, off);
        if (lane >= (uint32_t)off) up = mul29(up, u);
        if (lane + off < 64u) down = mul29(down, d);
    }
    // one inversion per wave: of the product of all its lanes
    Fr29 inv_total = one_rp;
    if (lane == 63u) inv_total = inv29_rp(up, unpack29<Fr29P>(c783_plain));
    inv_total = shfl29(inv_total, 63);
    // 1 / acc of this lane = 1 / total x (product of the lanes below) x (product of the lanes above)
    const Fr29 below = shfl_up29(up, 1), above = shfl_down29(down, 1);
    Fr29 iv = inv_total;
    if (lane > 0u) iv = mul29(iv, below);
    if (lane < 63u) iv = mul29(iv, above);
    const Fr29 one_r = unpack29<Fr29P>(Fr::one());                     // x 2^256 / 2^261: R' -> R on the way out
#pragma unroll
    for (int k = BI_K; k-- > 0;) {
        if (x[k].is_zero()) continue;
        const uint64_t i = (uint64_t)k * nt + t;
        stg(a + i, pack29_lt2p(mul29(mul29(iv, pre[k]), one_r)));
        iv = mul29(iv, v[k]);
    }
}

// ------------------------------------------------------------------------------------- scans
// Exclusive scan z[0] = id, z[i+1] = z[i] (op) a[i] in three kernels:
//   A: each block scans a contiguous segment of SCAN_BLOCK*SCAN_PER elements (thread-local chunk
//      + LDS Hillis-Steele over the 256 thread totals), writes locally-prefixed z, block total
//   B: one block scans the block totals
//   C: z[i] = off[block] (op) z[i]
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_PER = 8;
struct OpMul { __device__ static Fr id() { return Fr::one(); } __device__ static Fr f(const Fr& a, const Fr& b) { return a * b; } };
struct OpAdd { __device__ static Fr id() { return Fr::zero(); } __device__ static Fr f(const Fr& a, const Fr& b) { return a + b; } };

template <class Op>
__device__ __forceinline__ Fr block_exclusive_scan(Fr v, Fr* sh, Fr* total) {
    // sh: SCAN_THREADS entries.  Returns exclusive prefix of v over the block; *total = block sum.
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
        Fr x = sh[t];
        if (t >= off) x = Op::f(sh[t - off], x);
        __syncthreads();
        sh[t] = x;
        __syncthreads();
    }
    *total = sh[SCAN_THREADS - 1];
    Fr ex = t ? sh[t - 1] : Op::id();
    __syncthreads();
    return ex;
}

template <class Op>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_a(const Fr* __restrict__ a, Fr* __restrict__ z, Fr* __restrict__ totals, uint64_t n) {
    __shared__ Fr sh[SCAN_THREADS];
    const uint64_t seg = (uint64_t)blockIdx.x * (SCAN_THREADS * SCAN_PER);
    const uint64_t start = seg + (uint64_t)threadIdx.x * SCAN_PER;
    Fr loc[SCAN_PER];
    Fr acc = Op::id();
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) {
        const uint64_t i = start + k;
        loc[k] = acc;                       // exclusive within the thread
        if (i < n) acc = Op::f(acc, ldg(a + i));
    }
    Fr total;
    Fr ex = block_exclusive_scan<Op>(acc, sh, &total);
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) {
        const uint64_t i = start + k;
        if (i < n) stg(z + i, Op::f(ex, loc[k]));
    }
    if (threadIdx.x == 0) stg(totals + blockIdx.x, total);
}
// exclusive scan of `cnt` block totals in place, one block; handles cnt > SCAN_THREADS by a
// serial carry over SCAN_THREADS-sized windows
template <class Op>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_b(Fr* totals, uint32_t cnt) {
    __shared__ Fr sh[SCAN_THREADS];
    Fr carry = Op::id();
    for (uint32_t base = 0; base < cnt; base += SCAN_THREADS) {
        const uint32_t i = base + threadIdx.x;
        Fr v = i < cnt ? ldg(totals + i) : Op::id();
        Fr total;
        Fr ex = block_exclusive_scan<Op>(v, sh, &total);
        if (i < cnt) stg(totals + i, Op::f(carry, ex));
        carry = Op::f(carry, total);
    }
}
template <class Op>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_c(Fr* __restrict__ z, const Fr* __restrict__ offs, uint64_t n) {
    if (blockIdx.x == 0) return;   // offset of block 0 is the identity
    const Fr off = ldg(offs + blockIdx.x);
    const uint64_t seg = (uint64_t)blockIdx.x * (SCAN_THREADS * SCAN_PER);
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) {
        const uint64_t i = seg + (uint64_t)k * SCAN_THREADS + threadIdx.x;   // coalesced
        if (i < n) stg(z + i, Op::f(off, ldg(z + i)));
    }
}

template <class Op>
static int scan_run(zk_ctx* ctx, const Fr* d_a, Fr* d_z, uint64_t n) {
    if (!n) return ZK_OK;
    const uint32_t blocks = (uint32_t)((n + SCAN_THREADS * SCAN_PER - 1) / (SCAN_THREADS * SCAN_PER));
    Fr* totals = (Fr*)ctx->get_scratch(SC_POLY2, sizeof(Fr) * blocks);
    if (!totals) return ZK_ERR_OOM;
    hipLaunchKernelGGL((k_scan_a<Op>), dim3(blocks), dim3(SCAN_THREADS), 0, ctx->stream, d_a, d_z, totals, n);
    ZK_CHECK_LAUNCH(ctx);
    if (blocks > 1) {
        hipLaunchKernelGGL((k_scan_b<Op>), dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, totals, blocks);
        ZK_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL((k_scan_c<Op>), dim3(blocks), dim3(SCAN_THREADS), 0, ctx->stream, d_z, totals, n);
        ZK_CHECK_LAUNCH(ctx);
    }
    return ZK_OK;
}

// --------------------------------------------------------------- eval_polynomial / kate_division
// w[i] = c[i] * x^i through a two-level power table; block tree-sum; final sum on one block.
__device__ __forceinline__ Fr two_level_pow(const Fr* __restrict__ lo, const Fr* __restrict__ hi, int h, uint64_t e) {
    return ldg(lo + (e & ((1ull << h) - 1))) * ldg(hi + (e >> h));
}
__device__ __forceinline__ Fr block_sum(Fr v, Fr* sh) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int off = blockDim.x / 2; off > 0; off >>= 1) {
        if (t < off) sh[t] = sh[t] + sh[t + off];
        __syncthreads();
    }
    Fr r = sh[0];
    __syncthreads();
    return r;
}
__global__ void __launch_bounds__(256) k_eval_partial(const Fr* __restrict__ c, uint64_t n, const Fr* __restrict__ lo, const Fr* __restrict__ hi, int h, Fr* __restrict__ partial) {
    __shared__ Fr sh[256];
    Fr acc = Fr::zero();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        acc = acc + ldg(c + i) * two_level_pow(lo, hi, h, i);
    Fr s = block_sum(acc, sh);
    if (threadIdx.x == 0) stg(partial + blockIdx.x, s);
}
// batched forms: blockIdx.y selects the polynomial
__global__ void __launch_bounds__(256) k_eval_partial_multi(const Fr* const* __restrict__ polys, uint64_t n, const Fr* __restrict__ lo, const Fr* __restrict__ hi, int h, Fr* __restrict__ partial) {
    __shared__ Fr sh[256];
    const Fr* c = polys[blockIdx.y];
    Fr acc = Fr::zero();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        acc = acc + ldg(c + i) * two_level_pow(lo, hi, h, i);
    Fr s = block_sum(acc, sh);
    if (threadIdx.x == 0) stg(partial + (size_t)blockIdx.y * gridDim.x + blockIdx.x, s);
}
// Batched evaluation, one product per coefficient: block b owns the contiguous segment [b * seg, (b + 1) * seg) of
// polynomial blockIdx.y; thread t runs Horner over its elements t, t + 256, ... of the segment in descending order with
// the multiplier x^256 (coalesced: consecutive threads read consecutive coefficients), on 29-bit limbs with unsettled
// sums, then lifts its value by x^t, the block sum by x^(segment start).  The two-level table is only read three times
// per thread instead of once per coefficient (k_eval_partial_multi: two 8 x 32 products per coefficient).
__global__ void __launch_bounds__(256) k_eval_horner_multi(const Fr* const* __restrict__ polys, uint64_t n, uint64_t seg, const Fr* __restrict__ lo, const Fr* __restrict__ hi, int h,
                                                           Fr* __restrict__ partial) {
    __shared__ Fr sh[256];
    const Fr* c = polys[blockIdx.y];
    const uint64_t start = (uint64_t)blockIdx.x * seg, end = min(n, start + seg);
    auto rprime = [](Fr x) { for (int i = 0; i < 5; ++i) x = dbl(x); return x; };       // R -> R' = 2^261 form: x 32
    const Fr29 xs = unpack29<Fr29P>(rprime(two_level_pow(lo, hi, h, 256)));
    Fr29 acc = unpack29<Fr29P>(Fr::zero());
    const int64_t iters = start < end ? (int64_t)((end - start + 255) / 256) : 0;
    for (int64_t j = iters - 1; j >= 0; --j) {
        const uint64_t i = start + (uint64_t)j * 256 + threadIdx.x;
        // mul29 leaves a normalised value below 2p; adding a canonical coefficient keeps the limbs below 2^30 and the value
        // below 3p: a valid first operand of the next product
        acc = mul29(acc, xs);
        if (i < end) acc = add29(acc, unpack29<Fr29P>(ldg(c + i)));
    }
    const Fr mine = pack29_lt2p(mul29(acc, unpack29<Fr29P>(rprime(two_level_pow(lo, hi, h, threadIdx.x)))));
    Fr s = block_sum(mine, sh);
    if (threadIdx.x == 0) stg(partial + (size_t)blockIdx.y * gridDim.x + blockIdx.x, start < end ? s * two_level_pow(lo, hi, h, start) : Fr::zero());
}
__global__ void __launch_bounds__(256) k_sum_final_multi(const Fr* __restrict__ partial, uint32_t cnt, Fr* __restrict__ out) {
    __shared__ Fr sh[256];
    Fr acc = Fr::zero();
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) acc = acc + ldg(partial + (size_t)blockIdx.x * cnt + i);
    Fr s = block_sum(acc, sh);
    if (threadIdx.x == 0) stg(out + blockIdx.x, s);
}
// ---- several points in one pass (zk_poly_eval_pairs): the evaluations of a proof open ~15 distinct points at k = 18 (a column read at 13
// rotations), most of them for one or two polynomials; one batched call per point was a dozen launches and a host round trip each.
// tabs: per point a two-level table, nlo + nhi entries; blockIdx.y = point
__global__ void __launch_bounds__(256) k_pow_tables_multi(const Fr* __restrict__ xs, const Fr* __restrict__ steps, Fr* __restrict__ tabs, uint32_t nlo, uint32_t nhi) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nlo + nhi) return;
    Fr b = j < nlo ? ldg(xs + blockIdx.y) : ldg(steps + blockIdx.y), r = Fr::one();
    uint32_t e = j < nlo ? j : j - nlo;
    while (e) {
        if (e & 1) r = r * b;
        b = sqr(b);
        e >>= 1;
    }
    stg(tabs + (size_t)blockIdx.y * (nlo + nhi) + j, r);
}
// k_eval_horner_multi with a point per polynomial: blockIdx.y = pair (polys[y] at the point whose table is tabs + pidx[y] * stride)
__global__ void __launch_bounds__(256) k_eval_horner_pairs(const Fr* const* __restrict__ polys, const uint32_t* __restrict__ pidx, uint64_t n, uint64_t seg,
                                                           const Fr* __restrict__ tabs, uint32_t nlo, uint32_t stride, int h, Fr* __restrict__ partial) {
    __shared__ Fr sh[256];
    const Fr* c = polys[blockIdx.y];
    const Fr* lo = tabs + (size_t)pidx[blockIdx.y] * stride;
    const Fr* hi = lo + nlo;
    const uint64_t start = (uint64_t)blockIdx.x * seg, end = min(n, start + seg);
    auto rprime = [](Fr x) { for (int i = 0; i < 5; ++i) x = dbl(x); return x; };
    const Fr29 xs = unpack29<Fr29P>(rprime(two_level_pow(lo, hi, h, 256)));
    Fr29 acc = unpack29<Fr29P>(Fr::zero());
    const int64_t iters = start < end ? (int64_t)((end - start + 255) / 256) : 0;
    for (int64_t j = iters - 1; j >= 0; --j) {
        const uint64_t i = start + (uint64_t)j * 256 + threadIdx.x;
        acc = mul29(acc, xs);
        if (i < end) acc = add29(acc, unpack29<Fr29P>(ldg(c + i)));
    }
    const Fr mine = pack29_lt2p(mul29(acc, unpack29<Fr29P>(rprime(two_level_pow(lo, hi, h, threadIdx.x)))));
    Fr s = block_sum(mine, sh);
    if (threadIdx.x == 0) stg(partial + (size_t)blockIdx.y * gridDim.x + blockIdx.x, start < end ? s * two_level_pow(lo, hi, h, start) : Fr::zero());
}
__global__ void __launch_bounds__(256) k_sum_final(const Fr* __restrict__ partial, uint32_t cnt, Fr* out) {
    __shared__ Fr sh[256];
    Fr acc = Fr::zero();
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) acc = acc + ldg(partial + i);
    Fr s = block_sum(acc, sh);
    if (threadIdx.x == 0) stg(out, s);
}
// w[i] = c[i+1] * x^i  (shifted weights for kate division), i in [0, n-1)
__global__ void k_weight_shift(const Fr* __restrict__ c, uint64_t nm1, const Fr* __restrict__ lo, const Fr* __restrict__ hi, int h, Fr* __restrict__ w) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nm1) stg(w + i, ldg(c + i + 1) * two_level_pow(lo, hi, h, i));
}
// q[i] = (total - excl[i]) * xinv^i  where excl = exclusive prefix sum of w and total = sum w
__global__ void k_kate_finish(const Fr* __restrict__ excl, const Fr* __restrict__ total, uint64_t nm1, const Fr* __restrict__ lo, const Fr* __restrict__ hi, int h, Fr* __restrict__ q) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nm1) stg(q + i, (ldg(total) - ldg(excl + i)) * two_level_pow(lo, hi, h, i));
}
__global__ void k_copy_shift(const Fr* __restrict__ c, uint64_t nm1, Fr* __restrict__ q) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nm1) stg(q + i, ldg(c + i + 1));
}

// two-level table of x^e for e < n into scratch slot; returns lo pointer, hi = lo + 2^h
static int build_pow_table(zk_ctx* ctx, int slot, const Fr& x, uint64_t n, Fr** lo, Fr** hi, int* h_out) {
    int bits = 1;
    while ((1ull << bits) < n) ++bits;
    const int h = (bits + 1) / 2;
    const uint32_t nlo = 1u << h, nhi = 1u << (bits - h);
    Fr* tab = (Fr*)ctx->get_scratch(slot, sizeof(Fr) * ((size_t)nlo + nhi));
    if (!tab) return ZK_ERR_OOM;
    hipLaunchKernelGGL(k_powers, dim3((nlo + 255) / 256), dim3(256), 0, ctx->stream, x, Fr::one(), tab, nlo, 0);
    Fr step = x;
    for (int i = 0; i < h; ++i) step = sqr(step);
    hipLaunchKernelGGL(k_powers, dim3((nhi + 255) / 256), dim3(256), 0, ctx->stream, step, Fr::one(), tab + nlo, nhi, 0);
    ZK_CHECK_LAUNCH(ctx);
    *lo = tab; *hi = tab + nlo; *h_out = h;
    return ZK_OK;
}


// dst[i * stride + offset] = src[i] * s: interleaves one coset's values into the extended domain
__global__ void __launch_bounds__(256) k_scatter_scaled(const Fr* __restrict__ src, uint64_t n, Fr s, Fr* __restrict__ dst, uint64_t stride, uint64_t offset) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    stg(dst + i * stride + offset, ldg(src + i) * s);
}

// ---- uniformly random field elements: ChaCha20 block -> from_uniform_bytes ----------------------
// Element i is halo2curves' Fr::from_uniform_bytes (lo + hi * 2^256 mod r over the 64 little-endian
// bytes) of ChaCha20 block i: key = the caller's 32 bytes, 64-bit block counter = i in state words
// 12..13, 64-bit stream id in words 14..15 (the djb layout, which rand_chacha's ChaCha20Rng uses
// too).  Counter mode, so the blinding polynomial of a proof is one launch instead of n sequential
// draws from a host generator.
struct ChaChaKey { uint32_t w[8]; };
__device__ __forceinline__ uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
#define ZK_CHACHA_QR(a, b, c, d) a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7);
__global__ void __launch_bounds__(256) k_fr_random(ChaChaKey key, uint32_t s_lo, uint32_t s_hi, uint64_t first, Fr* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t ctr = first + i;
    const uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.w[0], key.w[1], key.w[2], key.w[3],
                             key.w[4], key.w[5], key.w[6], key.w[7], (uint32_t)ctr, (uint32_t)(ctr >> 32), s_lo, s_hi};
    uint32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6], x7 = in[7];
    uint32_t x8 = in[8], x9 = in[9], x10 = in[10], x11 = in[11], x12 = in[12], x13 = in[13], x14 = in[14], x15 = in[15];
    for (int r = 0; r < 10; ++r) {
        ZK_CHACHA_QR(x0, x4, x8, x12) ZK_CHACHA_QR(x1, x5, x9, x13) ZK_CHACHA_QR(x2, x6, x10, x14) ZK_CHACHA_QR(x3, x7, x11, x15)
        ZK_CHACHA_QR(x0, x5, x10, x15) ZK_CHACHA_QR(x1, x6, x11, x12) ZK_CHACHA_QR(x2, x7, x8, x13) ZK_CHACHA_QR(x3, x4, x9, x14)
    }
    Fr lo{{x0 + in[0], x1 + in[1], x2 + in[2], x3 + in[3], x4 + in[4], x5 + in[5], x6 + in[6], x7 + in[7]}};
    Fr hi{{x8 + in[8], x9 + in[9], x10 + in[10], x11 + in[11], x12 + in[12], x13 + in[13], x14 + in[14], x15 + in[15]}};
    // lo, hi may exceed r: they go in as the row operand of the CIOS product (any 256-bit value is fine there)
    stg(out + i, Fr::r2() * lo + (Fr::r2() * hi) * Fr::r2());
}
#undef ZK_CHACHA_QR

// dst[c][i] = src[c][i] + k for up to 64 columns per launch (the pointers travel as kernel arguments: no table upload, no host
// synchronisation).  The logUp sums add beta to every compressed table and input column of a proof before ONE batch inversion: as
// one two-instruction program per column through zk_quotient_eval that was ~400 uploads + stream synchronisations per proof.
struct AddConstBatch { const Fr* src[64]; Fr* dst[64]; };
__global__ void __launch_bounds__(256) k_add_const_many(AddConstBatch b, Fr k, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) stg(b.dst[blockIdx.y] + i, ldg(b.src[blockIdx.y] + i) + k);
}
int fr_add_const_many(zk_ctx* ctx, const void* const* d_src, void* const* d_dst, size_t count, const void* h_k, size_t n) {
    Fr k;
    memcpy(&k, h_k, sizeof k);
    for (size_t c0 = 0; c0 < count; c0 += 64) {
        AddConstBatch b;
        const size_t cnt = std::min<size_t>(64, count - c0);
        for (size_t c = 0; c < cnt; ++c) { b.src[c] = (const Fr*)d_src[c0 + c]; b.dst[c] = (Fr*)d_dst[c0 + c]; }
        for (size_t c = cnt; c < 64; ++c) { b.src[c] = nullptr; b.dst[c] = nullptr; }
        hipLaunchKernelGGL(k_add_const_many, dim3((unsigned)((n + 255) / 256), (unsigned)cnt), dim3(256), 0, ctx->stream, b, k, (uint64_t)n);
    }
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}
}  // namespace zk

using namespace zk;

extern "C" {

int zk_field_vec_op(zk_ctx* ctx, int field, int op, const void* d_a, const void* d_b, void* d_out, size_t n) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_a && d_b && d_out, "null device pointer");
    if (n == 0) return ZK_OK;
    if (field == ZK_FIELD_FR) return vec_op_launch<Fr>(ctx, op, d_a, d_b, d_out, n);
    if (field == ZK_FIELD_FQ) return vec_op_launch<Fq>(ctx, op, d_a, d_b, d_out, n);
    return ctx->fail(ZK_ERR_INVALID_ARG, "unknown field %d", field);
}

int zk_fr_scale(zk_ctx* ctx, void* d_a, const void* h_s, size_t n) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_a && h_s, "null pointer");
    return fr_scale_run(ctx, (Fr*)d_a, *(const Fr*)h_s, n);
}

int zk_fr_batch_invert(zk_ctx* ctx, void* d_a, size_t n) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_a, "null pointer");
    if (!n) return ZK_OK;
    uint64_t nt = (n + BI_K - 1) / BI_K;
    nt = (nt + 63) / 64 * 64;
    static const Fr c783 = [] { Fr c = Fr::one(); for (int i = 0; i < 783 - 256; ++i) c = dbl(c); return c; }();      // the integer 2^783 mod r (Fr::one() holds 2^256)
    hipLaunchKernelGGL(k_batch_invert, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, ctx->stream, (Fr*)d_a, (uint64_t)n, nt, c783);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}

int zk_fr_prefix_product(zk_ctx* ctx, const void* d_a, void* d_z, size_t n) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_a && d_z && d_a != d_z, "null or aliased pointer");
    return scan_run<OpMul>(ctx, (const Fr*)d_a, (Fr*)d_z, n);
}
int zk_fr_prefix_sum(zk_ctx* ctx, const void* d_a, void* d_z, size_t n) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_a && d_z && d_a != d_z, "null or aliased pointer");
    return scan_run<OpAdd>(ctx, (const Fr*)d_a, (Fr*)d_z, n);
}

int zk_poly_eval(zk_ctx* ctx, const void* d_coeffs, size_t n, const void* h_x, void* h_out) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_coeffs && h_x && h_out, "null pointer");
    if (n == 0) { memset(h_out, 0, sizeof(Fr)); return ZK_OK; }
    Fr *lo, *hi; int h;
    int rc = build_pow_table(ctx, SC_TMP, *(const Fr*)h_x, n, &lo, &hi, &h);
    if (rc) return rc;
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    Fr* partial = (Fr*)ctx->get_scratch(SC_POLY2, sizeof(Fr) * (blocks + 1));
    if (!partial) return ZK_ERR_OOM;
    hipLaunchKernelGGL(k_eval_partial, dim3(blocks), dim3(256), 0, ctx->stream, (const Fr*)d_coeffs, (uint64_t)n, lo, hi, h, partial);
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, ctx->stream, partial, blocks, partial + blocks);
    ZK_CHECK_LAUNCH(ctx);
    ZK_HIP(ctx, hipMemcpyAsync(h_out, partial + blocks, sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

// eval_polynomial of `count` polynomials at ONE point: one power table, one launch pair, one sync
// (the prover opens hundreds of columns at the same few points x * omega^rot)
int zk_poly_eval_batch(zk_ctx* ctx, const void* const* d_coeff_ptrs, size_t count, size_t n, const void* h_x, void* h_out) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, (d_coeff_ptrs || !count) && h_x && h_out, "null pointer");
    if (count == 0) return ZK_OK;
    if (n == 0) { memset(h_out, 0, sizeof(Fr) * count); return ZK_OK; }
    Fr *lo, *hi; int h;
    int rc = build_pow_table(ctx, SC_TMP, *(const Fr*)h_x, n, &lo, &hi, &h);
    if (rc) return rc;
    // segments of >= 4096 coefficients (16 Horner steps per thread) where the polynomial has them: a thread pays three table look-ups
    // and two lifting products whatever its share, and at 2^18 a 1024-coefficient segment made those the larger half of its work
    uint32_t blocks = n >= 512 ? (uint32_t)((n + 4095) / 4096) : (uint32_t)((n + 255) / 256);
    if (blocks > 256) blocks = 256;
    char* sc = (char*)ctx->get_scratch(SC_POLY2, sizeof(Fr) * ((size_t)blocks + 1) * count + 8 * count + 64);
    if (!sc) return ZK_ERR_OOM;
    Fr* partial = (Fr*)sc;
    Fr* results = partial + (size_t)blocks * count;
    const Fr** d_ptrs = (const Fr**)(results + count);
    ZK_HIP(ctx, hipMemcpyAsync(d_ptrs, d_coeff_ptrs, 8 * count, hipMemcpyHostToDevice, ctx->stream));
    if (n >= 512) {
        const uint64_t seg = (((uint64_t)n + blocks - 1) / blocks + 255) & ~(uint64_t)255;        // whole strides of 256 per segment
        hipLaunchKernelGGL(k_eval_horner_multi, dim3(blocks, (unsigned)count), dim3(256), 0, ctx->stream, (const Fr* const*)d_ptrs, (uint64_t)n, seg, lo, hi, h, partial);
    } else {
        hipLaunchKernelGGL(k_eval_partial_multi, dim3(blocks, (unsigned)count), dim3(256), 0, ctx->stream, (const Fr* const*)d_ptrs, (uint64_t)n, lo, hi, h, partial);
    }
    hipLaunchKernelGGL(k_sum_final_multi, dim3((unsigned)count), dim3(256), 0, ctx->stream, (const Fr*)partial, blocks, results);
    ZK_CHECK_LAUNCH(ctx);
    ZK_HIP(ctx, hipMemcpyAsync(h_out, results, sizeof(Fr) * count, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

int zk_poly_eval_pairs(zk_ctx* ctx, const void* const* d_coeff_ptrs, const uint32_t* point_index, size_t count, const void* h_points, size_t num_points, size_t n, void* h_out) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, (d_coeff_ptrs && point_index || !count) && (h_points || !num_points) && (h_out || !count), "null pointer");
    if (count == 0) return ZK_OK;
    for (size_t j = 0; j < count; ++j) ZK_REQUIRE(ctx, point_index[j] < num_points, "point index out of range");
    if (n == 0) { memset(h_out, 0, sizeof(Fr) * count); return ZK_OK; }
    if (n < 512 || num_points > 4096 || count > 65535) {
        // short polynomials (the single-product kernel has no Horner stride to fill) and oversized batches: point by point
        const Fr* pts = (const Fr*)h_points;
        for (size_t p = 0; p < num_points; ++p) {
            std::vector<const void*> ptrs;
            std::vector<size_t> where;
            for (size_t j = 0; j < count; ++j) if (point_index[j] == p) { ptrs.push_back(d_coeff_ptrs[j]); where.push_back(j); }
            if (ptrs.empty()) continue;
            std::vector<Fr> vals(ptrs.size());
            int rc = zk_poly_eval_batch(ctx, ptrs.data(), ptrs.size(), n, pts + p, vals.data());
            if (rc) return rc;
            for (size_t t = 0; t < where.size(); ++t) ((Fr*)h_out)[where[t]] = vals[t];
        }
        return ZK_OK;
    }
    int bits = 1;
    while ((1ull << bits) < n) ++bits;
    const int h = (bits + 1) / 2;
    const uint32_t nlo = 1u << h, nhi = 1u << (bits - h), stride = nlo + nhi;
    uint32_t blocks = (uint32_t)((n + 4095) / 4096);          // >= 16 Horner steps per thread (see zk_poly_eval_batch)
    if (blocks > 256) blocks = 256;
    // scratch: [tables P x stride][partials count x blocks][results count][points P][steps P][polys count][pidx count]
    const size_t fr_words = (size_t)num_points * stride + (size_t)blocks * count + count + 2 * num_points;
    char* sc = (char*)ctx->get_scratch(SC_POLY2, sizeof(Fr) * fr_words + 12 * count + 64);
    if (!sc) return ZK_ERR_OOM;
    Fr* tabs = (Fr*)sc;
    Fr* partial = tabs + (size_t)num_points * stride;
    Fr* results = partial + (size_t)blocks * count;
    Fr* d_xs = results + count;
    Fr* d_steps = d_xs + num_points;
    const Fr** d_ptrs = (const Fr**)(d_steps + num_points);
    uint32_t* d_pidx = (uint32_t*)(d_ptrs + count);
    std::vector<Fr> steps(num_points);
    for (size_t p = 0; p < num_points; ++p) {
        Fr st = ((const Fr*)h_points)[p];
        for (int i = 0; i < h; ++i) st = sqr(st);
        steps[p] = st;
    }
    ZK_HIP(ctx, hipMemcpyAsync(d_xs, h_points, sizeof(Fr) * num_points, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipMemcpyAsync(d_steps, steps.data(), sizeof(Fr) * num_points, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipMemcpyAsync(d_ptrs, d_coeff_ptrs, 8 * count, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipMemcpyAsync(d_pidx, point_index, 4 * count, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_pow_tables_multi, dim3((stride + 255) / 256, (unsigned)num_points), dim3(256), 0, ctx->stream, (const Fr*)d_xs, (const Fr*)d_steps, tabs, nlo, nhi);
    const uint64_t seg = (((uint64_t)n + blocks - 1) / blocks + 255) & ~(uint64_t)255;
    hipLaunchKernelGGL(k_eval_horner_pairs, dim3(blocks, (unsigned)count), dim3(256), 0, ctx->stream, (const Fr* const*)d_ptrs, (const uint32_t*)d_pidx, (uint64_t)n, seg,
                       (const Fr*)tabs, nlo, stride, h, partial);
    hipLaunchKernelGGL(k_sum_final_multi, dim3((unsigned)count), dim3(256), 0, ctx->stream, (const Fr*)partial, blocks, results);
    ZK_CHECK_LAUNCH(ctx);
    ZK_HIP(ctx, hipMemcpyAsync(h_out, results, sizeof(Fr) * count, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));          // also covers the staging vector `steps`
    return ZK_OK;
}

int zk_kate_division(zk_ctx* ctx, const void* d_coeffs, size_t n, const void* h_z, void* d_q) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_coeffs && h_z && d_q && d_q != d_coeffs, "null or aliased pointer");
    if (n < 2) return ZK_OK;
    const uint64_t nm1 = n - 1;
    const Fr z = *(const Fr*)h_z;
    const dim3 g((unsigned)((nm1 + 255) / 256)), t(256);
    if (z.is_zero()) {   // q[i] = c[i+1]
        hipLaunchKernelGGL(k_copy_shift, g, t, 0, ctx->stream, (const Fr*)d_coeffs, nm1, (Fr*)d_q);
        ZK_CHECK_LAUNCH(ctx);
        return ZK_OK;
    }
    // q[i] = z^-i * sum_{j >= i} c[j+1] z^j  = z^-i * (total - exclusive_prefix(w)[i]),  w[j] = c[j+1] z^j
    Fr *lo, *hi; int h;
    int rc = build_pow_table(ctx, SC_TMP, z, nm1, &lo, &hi, &h);
    if (rc) return rc;
    Fr* w = (Fr*)ctx->get_scratch(SC_POLY, sizeof(Fr) * (2 * nm1 + 2));
    if (!w) return ZK_ERR_OOM;
    Fr* excl = w + nm1;
    Fr* total = excl + nm1;
    hipLaunchKernelGGL(k_weight_shift, g, t, 0, ctx->stream, (const Fr*)d_coeffs, nm1, lo, hi, h, w);
    ZK_CHECK_LAUNCH(ctx);
    rc = scan_run<OpAdd>(ctx, w, excl, nm1);
    if (rc) return rc;
    // total = excl[nm1-1] + w[nm1-1]
    hipLaunchKernelGGL((k_vec_op<Fr, ZK_OP_ADD>), dim3(1), dim3(64), 0, ctx->stream, (const Fr*)(excl + (nm1 - 1)), (const Fr*)(w + (nm1 - 1)), total, (uint64_t)1);
    ZK_CHECK_LAUNCH(ctx);
    Fr zinv = inv(z);
    rc = build_pow_table(ctx, SC_TMP, zinv, nm1, &lo, &hi, &h);
    if (rc) return rc;
    hipLaunchKernelGGL(k_kate_finish, g, t, 0, ctx->stream, (const Fr*)excl, (const Fr*)total, nm1, lo, hi, h, (Fr*)d_q);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}

int zk_fr_random(zk_ctx* ctx, const uint8_t* key32, uint64_t stream_id, uint64_t first_block, void* d_out, size_t n) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, key32 && (d_out || !n), "null pointer");
    if (!n) return ZK_OK;
    ChaChaKey key;
    memcpy(key.w, key32, 32);   // little-endian words
    hipLaunchKernelGGL(k_fr_random, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, key, (uint32_t)stream_id, (uint32_t)(stream_id >> 32), first_block, (Fr*)d_out, (uint64_t)n);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}

int zk_fr_scatter_scaled(zk_ctx* ctx, const void* d_src, size_t n, const void* h_scale, void* d_dst, size_t stride, size_t offset) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, (d_src && d_dst && h_scale) || !n, "null pointer");
    ZK_REQUIRE(ctx, stride >= 1 && offset < stride, "need offset < stride");
    if (!n) return ZK_OK;
    hipLaunchKernelGGL(k_scatter_scaled, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const Fr*)d_src, (uint64_t)n, *(const Fr*)h_scale, (Fr*)d_dst, (uint64_t)stride, (uint64_t)offset);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}

// ---- self-test of the device product routines -----------------------------------------------------------------------------------
// The Montgomery products run as ONE asm statement each on the device (csrc/mul29_asm.hip.hpp, generated); the C forms of
// ff29.hip.hpp stay the definition, are what the host build runs (tests/test_host_arith.py) and what the asm forms must equal bit
// for bit at the lazy-reduction bounds every kernel relies on.  One lane = one operand set: limbs of the first operand below 2^30
// (value below 2^258), second operand normalised, the addends of mul2add29 at their bounds; every sixteenth lane holds every
// limb AT its bound.  Both fields (Fr: NTT / evaluator, Fq: the bucket additions).
} // extern "C"
namespace zk {
__device__ __forceinline__ uint32_t st_hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <class P>
__global__ void k_selftest_products(uint32_t* bad, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    F29<P> a, b, c, d, u;
    for (int i = 0; i < 9; ++i) {
        const uint32_t top = i == 8;
        a.l[i] = st_hash32(seed + tid * 41 + i) & (top ? 0x3ffffffu : 0x3fffffffu);          // limbs < 2^30, value < 2^258
        b.l[i] = st_hash32(seed + tid * 43 + i + 100) & (top ? 0xffffffu : 0x1fffffffu);     // normalised, < 2^256
        c.l[i] = st_hash32(seed + tid * 47 + i + 200) & (top ? 0xffffffu : 0x1fffffffu);
        d.l[i] = st_hash32(seed + tid * 53 + i + 300) & (top ? 0xffffffu : 0x3fffffffu);     // limbs < 2^30
        u.l[i] = st_hash32(seed + blockIdx.x * 59 + i + 400) & (top ? 0xffffffu : 0x1fffffffu);   // workgroup-uniform second factor (scalar registers)
        if ((tid & 15) == 3 && i < 8) { a.l[i] = 0x3fffffffu; b.l[i] = 0x1fffffffu; c.l[i] = 0x1fffffffu; d.l[i] = 0x3fffffffu; }
    }
    F29<P> an = a;
    for (int i = 0; i < 8; ++i) an.l[i] &= 0x1fffffffu;
    uint32_t diff = 0;
    auto cmp = [&](const F29<P>& x, const F29<P>& y, uint32_t bit) { for (int i = 0; i < 9; ++i) if (x.l[i] != y.l[i]) diff |= bit; };
    cmp(mul29(a, b), mul29_c(a, b), 1u);
    cmp(mul29_ub(a, u), mul29_ub_c(a, u), 2u);
    cmp(sqr29(an), sqr29_c(an), 4u);
    cmp(mul2add29(an, b, c, d), mul2add29_c(an, b, c, d), 8u);
    // and the product against the definition: a b 2^-261 mod p through the 8 x 32 CIOS routine is covered by the NTT / MSM parity
    // tests; here the two forms of the SAME column sums must agree in every limb
    if (diff) { atomicAdd(bad, 1u); atomicOr(bad + 1, diff); }
    atomicAdd(bad + 2, 1u);
}
}  // namespace zk
extern "C" {
int zk_selftest_products(zk_ctx* ctx, int field, uint32_t operand_sets, uint32_t seed, uint32_t* out3) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, out3 && (field == ZK_FIELD_FR || field == ZK_FIELD_FQ), "bad argument");
    uint32_t* d = (uint32_t*)ctx->get_scratch(SC_TMP, 64);
    if (!d) return ZK_ERR_OOM;
    ZK_HIP(ctx, hipMemsetAsync(d, 0, 12, ctx->stream));
    const unsigned blocks = (operand_sets + 255) / 256;
    if (blocks) {
        if (field == ZK_FIELD_FR) hipLaunchKernelGGL(zk::k_selftest_products<Fr29P>, dim3(blocks), dim3(256), 0, ctx->stream, d, seed);
        else hipLaunchKernelGGL(zk::k_selftest_products<Fq29P>, dim3(blocks), dim3(256), 0, ctx->stream, d, seed);
        ZK_CHECK_LAUNCH(ctx);
    }
    ZK_HIP(ctx, hipMemcpyAsync(out3, d, 12, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}
}  // extern "C"
