// BN254 Fr NTT for gfx950: halo2_proofs::arithmetic::best_fft / poly::EvaluationDomain
// (external crate; reference call sites A1-A4).
//
// Definition (natural order in -> natural order out):  out[i] = sum_j a[j] * omega^(i*j).
//
// MI355X design: mixed-radix Cooley-Tukey in P <= 3 passes of <= 2^10 points each.  A pass stages
// a [digit x T] tile in LDS (T consecutive elements of the fastest-varying remaining index, so
// every global access is a T*32-byte contiguous run), runs all log2(n_p) radix-2 stages out of
// LDS, applies the inter-pass twiddle omega^(j''*i_p) on the way out (two-level table) and writes
// back.  Input index is read big-endian in the digits, output little-endian; the last pass writes
// to the digit-reversed position, so no separate transpose / bit-reversal kernel exists.
//
// Arithmetic: inside a pass elements live in LDS as nine 29-bit limbs (ff29.hip.hpp; one u32 plane
// per limb, so a wave's accesses are 4-byte strided and bank-conflict-free on contiguous runs).
// Butterflies use the carry-free 29-bit Montgomery product with twiddles held in R' = 2^261 form,
// so data stays in halo2curves' R = 2^256 form with no conversion; sums are kept lazily reduced
// and only the value written back to HBM is brought to the canonical representative.
#include "ctx.hpp"
#include "ff29.hip.hpp"

namespace zk {

constexpr int NTT_MAX_DIGIT = 10;
constexpr int NTT_TILE = 4096;         // elements staged per workgroup (9 x 4 B x 4096 = 144 KiB LDS)
constexpr int NTT_THREADS = 1024;
constexpr int NTT_LDS_BYTES_PER_ELT = 36;

struct Tw29;        // one butterfly twiddle split into 29-bit limbs (48 B), defined with the kernels

// One launch transforms up to NTT_BATCH columns over the same domain: blockIdx.y selects the column.  A 2^18 column
// is 64 tiles -- a quarter of the CUs --, and a prover transforms hundreds of columns per stage.
constexpr int NTT_BATCH = 16;
struct NttIo { const Fr* src[NTT_BATCH]; Fr* dst[NTT_BATCH]; };

struct NttPass {
    int log_np;     // digit size
    int log_m;      // stride of the digit (non-last)
    const Tw29* tw; // n_p/2 butterfly twiddles (omega^(n/n_p))^x, R' form, 29-bit limbs
    const Fr* out_tw = nullptr;   // non-last passes: inter-pass twiddle of every output element, in output order (R' form)
};

struct NttDomain {
    uint32_t log_n = 0;
    int npass = 0;
    NttPass pass[3];
    int h = 0;                // two-level split: omega^e = lo[e & (2^h-1)] * hi[e >> h]
    Fr* d_lo = nullptr;       // 2^h entries, R' form
    Fr* d_hi = nullptr;       // 2^(log_n-h) entries, R' form
    Fr* d_tw[3] = {nullptr, nullptr, nullptr};
    Tw29* d_tw29[3] = {nullptr, nullptr, nullptr};
    Fr* d_out_tw[2] = {nullptr, nullptr};
    Fr final_mul;             // (scale or 1) in R' form: last-pass output multiplier
    bool fin_folded = false;  // final_mul is already part of the last inter-pass twiddle table: the last pass only reduces
    ~NttDomain() {
        if (d_lo) (void)hipFree(d_lo);
        if (d_hi) (void)hipFree(d_hi);
        for (auto p : d_tw) if (p) (void)hipFree(p);
        for (auto p : d_tw29) if (p) (void)hipFree((void*)p);
        for (auto p : d_out_tw) if (p) (void)hipFree(p);
    }
};

// ------------------------------------------------------------------------------------- helpers
__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// R (2^256) Montgomery form -> R' (2^261) form: multiply by 32
__host__ __device__ __forceinline__ Fr to_rprime(Fr x) {
#pragma unroll
    for (int i = 0; i < 5; ++i) x = dbl(x);
    return x;
}

struct Lds29 {
    uint32_t* p;     // 9 planes of `stride` u32
    int stride;
    __device__ __forceinline__ Fr29 load(int idx) const {
        Fr29 r;
#pragma unroll
        for (int k = 0; k < 9; ++k) r.l[k] = p[k * stride + idx];
        return r;
    }
    __device__ __forceinline__ void store(int idx, const Fr29& v) const {
#pragma unroll
        for (int k = 0; k < 9; ++k) p[k * stride + idx] = v.l[k];
    }
};

// out[j] = base^j * mul   (table builder; one thread per entry, square-and-multiply).
// rprime != 0: store in R' = 2^261 Montgomery form.
__global__ void k_powers(Fr base, Fr mul, Fr* out, uint32_t count, int rprime) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fr r = mul, b = base;
    uint32_t e = j;
    while (e) {
        if (e & 1) r = r * b;
        b = sqr(b);
        e >>= 1;
    }
    stg(out + j, rprime ? to_rprime(r) : r);
}

__device__ __forceinline__ Fr two_level(const Fr* __restrict__ lo, const Fr* __restrict__ hi, int h, uint32_t e) {
    return ldg(lo + (e & ((1u << h) - 1))) * ldg(hi + (e >> h));
}
// both tables in R' form -> product in R' form, normalised, < 2p
__device__ __forceinline__ Fr29 two_level29(const Fr* __restrict__ lo, const Fr* __restrict__ hi, int h, uint32_t e) {
    return mul29(unpack29<Fr29P>(ldg(lo + (e & ((1u << h) - 1)))), unpack29<Fr29P>(ldg(hi + (e >> h))));
}

// out_tw[base + d*m + c] = omega^(((blk*T + c) * d) << tw_shift): the twiddle a non-last pass applies to
// each element it writes, tabulated once per domain in output order
__global__ void k_build_out_twiddles(Fr* __restrict__ out_tw, const Fr* __restrict__ lo, const Fr* __restrict__ hi, int h, int log_np, int log_m, int tw_shift, uint64_t n,
                                     Fr fin, int apply_fin) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint64_t m = 1ull << log_m;
    const uint32_t jpp = (uint32_t)(g & (m - 1)), d = (uint32_t)(g >> log_m) & ((1u << log_np) - 1);
    Fr29 v = two_level29(lo, hi, h, (jpp * d) << tw_shift);
    if (apply_fin) v = mul29(v, unpack29<Fr29P>(fin));      // the transform's output scale rides on the last inter-pass twiddle
    stg(out_tw + g, pack29_lt2p(v));
}

// ------------------------------------------------------------------------------- butterflies
// (u, x) -> (u + x w, u - x w); x w is reduced below 2p by the product, the sums stay lazy
__device__ __forceinline__ void bfly29(Fr29& u, Fr29& x, const Fr29& w) {
    const Fr29 v = mul29(x, w);
    Fr29 a0 = add29(u, v), a1 = sub29k<4>(u, v);
    normalize29(a0);
    normalize29(a1);
    u = a0;
    x = a1;
}
// butterfly twiddles are tabulated already split into 29-bit limbs (12 words = 48 B per entry, three
// 16-byte loads): no 8 x 32 -> 9 x 29 repacking in front of every product
struct Tw29 { uint4 q[3]; };
__device__ __forceinline__ Fr29 tw29(const Tw29* __restrict__ tw, uint32_t idx) {
    const uint4 a = tw[idx].q[0], b = tw[idx].q[1], c = tw[idx].q[2];
    return Fr29{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x}};
}
__global__ void k_split_twiddles(const Fr* __restrict__ in, Tw29* __restrict__ out, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const Fr29 v = unpack29<Fr29P>(ldg(in + i));
    out[i].q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    out[i].q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    out[i].q[2] = make_uint4(v.l[8], 0u, 0u, 0u);
}
// DIT stages s (and s+1 when R == 2) on the 2^R elements at digit offsets {0, h, 2h, 3h}, h = 2^s,
// j = digit mod h.  Radix-4 keeps both stages in registers: half the LDS round trips and barriers
// of two radix-2 stages, same four products.
// Inside a step the sums are not carry-propagated between the two stages.  Limb bounds with
// N = 2^29 (operands enter normalised, limbs < N; x w is normalised and < 1.4 p; the balanced
// limbs of 2p are < 2N):  stage 1 leaves u + v < 2N and u + 2p - v < 3N; stage 2 multiplies such
// an operand -- mul29 takes limbs up to 2^31.2 (9 N (X + N) < 2^64) -- and leaves sums < 5N < 2^32.
// One normalisation per element at the end of the step instead of one per butterfly output.
__device__ __forceinline__ void bfly29_lazy(Fr29& u, Fr29& x, const Fr29& w) {
    const Fr29 v = mul29(x, w);
    const Fr29 a0 = add29(u, v), a1 = sub29k<2>(u, v);
    u = a0;
    x = a1;
}
template <int R>
__device__ __forceinline__ void dit_step(Fr29 (&e)[4], const Tw29* __restrict__ tw, int log_np, int s, int j) {
    const Fr29 w0 = tw29(tw, (uint32_t)j << (log_np - 1 - s));
    bfly29_lazy(e[0], e[1], w0);
    if (R == 2) {
        bfly29_lazy(e[2], e[3], w0);
        bfly29_lazy(e[0], e[2], tw29(tw, (uint32_t)j << (log_np - 2 - s)));
        bfly29_lazy(e[1], e[3], tw29(tw, (uint32_t)(j + (1 << s)) << (log_np - 2 - s)));
        normalize29(e[2]);
        normalize29(e[3]);
    }
    normalize29(e[0]);
    normalize29(e[1]);
}
// (u, x) -> (u + x, u - x) for reduced operands (< 2p each, or sums of two such: x < 4p)
__device__ __forceinline__ void bfly29_one(Fr29& u, Fr29& x) {
    Fr29 a0 = add29(u, x), a1 = sub29k<4>(u, x);
    normalize29(a0);
    normalize29(a1);
    u = a0;
    x = a1;
}
// First step of a pass (s = 0): its operands come reduced from global memory and every twiddle of
// stage 0 is omega^0 = 1, as is the j = 0 twiddle of stage 1 -- three of the four products of a
// radix-4 step (the one of a radix-2 step) are products by one and are skipped.
template <int R>
__device__ __forceinline__ void dit_first_step(Fr29 (&e)[4], const Tw29* __restrict__ tw, int log_np) {
    bfly29_one(e[0], e[1]);
    if (R == 2) {
        bfly29_one(e[2], e[3]);
        bfly29_one(e[0], e[2]);
        bfly29(e[1], e[3], tw29(tw, 1u << (log_np - 2)));
    }
}

// ------------------------------------------------------------------------------ non-last pass
// Tile = [n_p digits][T columns], element (d, c) lives at base + d*m + c with
// base = hi_idx * (n_p*m) + blk*T.  DIT: the first step reads its operands straight from global
// memory (digit bit-reversed), the steps in between go through LDS, the last step multiplies by
// the inter-pass twiddle omega^((j'' * i_p) << tw_shift), j'' = blk*T + c, and writes to global
// memory: no staging copy on either side.  Steps are radix-4 (radix-2 first when log_np is odd).
// LDS invariant: limbs 0..7 < 2^29 (normalised), value < 2^261.
__global__ void __launch_bounds__(NTT_THREADS)
k_ntt_pass(NttIo io, const Tw29* __restrict__ tw, const Fr* __restrict__ lo,
           const Fr* __restrict__ hi, int h, int log_np, int log_t, int log_m, int tw_shift, const Fr* __restrict__ pre,
           const Fr* __restrict__ out_tw, uint32_t ncols, int xcd_cols) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // One-dimensional grid over (tile, column).  The per-element tables (inter-pass twiddles, coset shifts: 32 B per element of the
    // TILE, the same for every column) are read by every column's workgroup of a tile: those workgroups are numbered so that they
    // are consecutive workgroups of ONE XCD (workgroups go to the eight XCDs round-robin, each XCD has its own L2) -- the table
    // lines are fetched from HBM once per tile instead of once per (tile, column).
    uint32_t tile_id, col;
    if (xcd_cols) { const uint32_t slot = blockIdx.x >> 3; col = slot % ncols; tile_id = (slot / ncols) * 8u + (blockIdx.x & 7u); }
    else { col = blockIdx.x % ncols; tile_id = blockIdx.x / ncols; }
    const Fr* __restrict__ src = io.src[col];
    Fr* __restrict__ dst = io.dst[col];
    const int tile = 1 << (log_np + log_t);
    Lds29 L{smem, tile};
    const int T = 1 << log_t;
    const uint64_t m = 1ull << log_m;
    const uint32_t tiles_per_hi = (uint32_t)(m >> log_t);
    const uint32_t hi_idx = tile_id / tiles_per_hi, blk = tile_id % tiles_per_hi;
    const uint64_t base = ((uint64_t)hi_idx << (log_np + log_m)) + ((uint64_t)blk << log_t);

    for (int s = 0; s < log_np;) {
        const int r = ((log_np - s) & 1) ? 1 : 2;
        const bool first = s == 0, last = s + r == log_np;
        const int hgt = 1 << s, items = tile >> r;
        for (int it = threadIdx.x; it < items; it += blockDim.x) {
            const int c = it & (T - 1), b = it >> log_t;      // c fastest: T-element contiguous runs in global memory
            const int j = b & (hgt - 1);
            const int lo_d = ((b >> s) << (s + r)) | j;
            Fr29 e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k >= (1 << r)) break;
                const int dl = lo_d + k * hgt;
                if (first) {
                    const uint64_t gi = base + (uint64_t)bitrev(dl, log_np) * m + c;
                    e[k] = unpack29<Fr29P>(ldg(src + gi));
                    if (pre) e[k] = mul29(e[k], unpack29<Fr29P>(ldg(pre + gi)));     // coset shift a[i] * g^i fused into the load (table in R' form)
                } else {
                    e[k] = L.load((dl << log_t) | c);
                }
            }
            if (first) { if (r == 2) dit_first_step<2>(e, tw, log_np); else dit_first_step<1>(e, tw, log_np); }
            else if (r == 2) dit_step<2>(e, tw, log_np, s, j);
            else dit_step<1>(e, tw, log_np, s, j);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k >= (1 << r)) break;
                const int dl = lo_d + k * hgt;
                if (last) {
                    const uint32_t jpp = (blk << log_t) + c;
                    const uint64_t go = base + (uint64_t)dl * m + c;
                    // inter-pass twiddle: one load from the per-domain table in output order (one product),
                    // or two table entries and two products when the table was not built
                    const Fr29 v = mul29(e[k], out_tw ? unpack29<Fr29P>(ldg(out_tw + go)) : two_level29(lo, hi, h, (jpp * (uint32_t)dl) << tw_shift));
                    stg(dst + go, pack29_raw(v));           // an intermediate: below 2p, not canonical (the next pass does not need it to be)
                } else {
                    L.store((dl << log_t) | c, e[k]);
                }
            }
        }
        s += r;
        if (s < log_np) __syncthreads();
    }
}

static bool ntt_xcd_remap() { static const bool on = getenv("ZK_NTT_XCD") && atoi(getenv("ZK_NTT_XCD")) == 1; return on; }      // measurement knob, off: neutral at every size (profiles/r03_ntt_xcd.md)
__host__ __device__ __forceinline__ int ntt_row_pad(int log_np) { return log_np >= 8 ? 8 : 0; }     // <= 16 rows per tile then: at most 128 extra elements
// ----------------------------------------------------------------------------------- last pass
// Rows of n_P contiguous elements; tile = T rows i1 = blk*T + c (row stride = midN * n_P) at a
// fixed middle digit `mid`.  LDS layout [c][d].  Same step structure as the other passes: the
// first step reads the bit-reversed digits of its rows from global memory (32-byte sectors of a
// 32 KiB row, all consumed by this workgroup in the same sweep), the last step writes
// output index = i1 + n1 * (mid + midN * i_P) with c fastest (T consecutive outputs).  Every
// output is multiplied by `fin` (1 or the inverse-transform scale, R' form), which also brings
// the lazy sums back below 2p.
__global__ void __launch_bounds__(NTT_THREADS)
k_ntt_last(NttIo io, const Tw29* __restrict__ tw, int log_np, int log_t,
           int log_n1, int log_mid, Fr fin, const Fr* __restrict__ pre, int fin_folded, int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const Fr* __restrict__ src = io.src[blockIdx.y];
    Fr* __restrict__ dst = io.dst[blockIdx.y];
    const int tile = 1 << (log_np + log_t);
    const int T = 1 << log_t;
    // rows are padded by 8 words: the last step reads with c fastest (coalesced output), and with a row
    // stride that is a multiple of the 32 banks the T rows of a wave would collide on every bank
    const int row = (1 << log_np) + ntt_row_pad(log_np);
    Lds29 L{smem, T * row};
    // Workgroups go to the eight XCDs round-robin and each XCD has its own L2.  A tile of T rows writes T * 32-byte runs
    // (64 B at T = 2), i.e. HALF of every 128-byte line it touches; the other half belongs to the tile next to it.  With the
    // plain numbering those two tiles run on different XCDs and each L2 writes its half back on its own (masked partial
    // writes); renumbered so that neighbouring tiles are consecutive workgroups of ONE XCD, the halves meet in that L2.
    uint32_t bx = blockIdx.x;
    if (xcd_remap) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    const uint32_t mid = bx & ((1u << log_mid) - 1), blk = bx >> log_mid;
    const Fr29 fin29 = unpack29<Fr29P>(fin);

    for (int s = 0; s < log_np;) {
        const int r = ((log_np - s) & 1) ? 1 : 2;
        const bool first = s == 0, last = s + r == log_np;
        const int hgt = 1 << s, items = tile >> r;
        for (int it = threadIdx.x; it < items; it += blockDim.x) {
            int c, b;
            if (last) { c = it & (T - 1); b = it >> log_t; }                              // c fastest: coalesced output
            else { b = it & ((1 << (log_np - r)) - 1); c = it >> (log_np - r); }          // d fastest: conflict-free LDS
            const int j = b & (hgt - 1);
            const int lo_d = ((b >> s) << (s + r)) | j;
            const uint64_t i1 = ((uint64_t)blk << log_t) + c;
            Fr29 e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k >= (1 << r)) break;
                const int dl = lo_d + k * hgt;
                if (first) {
                    const uint64_t gi = (((i1 << log_mid) + mid) << log_np) + bitrev(dl, log_np);
                    e[k] = unpack29<Fr29P>(ldg(src + gi));
                    if (pre) e[k] = mul29(e[k], unpack29<Fr29P>(ldg(pre + gi)));     // only when this is the only pass
                } else {
                    e[k] = L.load(c * row + dl);
                }
            }
            if (first) { if (r == 2) dit_first_step<2>(e, tw, log_np); else dit_first_step<1>(e, tw, log_np); }
            else if (r == 2) dit_step<2>(e, tw, log_np, s, j);
            else dit_step<1>(e, tw, log_np, s, j);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k >= (1 << r)) break;
                const int dl = lo_d + k * hgt;
                if (last) {
                    const uint64_t o = i1 + (((uint64_t)mid + ((uint64_t)dl << log_mid)) << log_n1);
                    stg(dst + o, fin_folded ? reduce_lazy29(e[k]) : pack29_lt2p(mul29(e[k], fin29)));
                } else {
                    L.store(c * row + dl, e[k]);
                }
            }
        }
        s += r;
        if (s < log_np) __syncthreads();
    }
}


// ------------------------------------------------------------------ the two passes with their step structure fixed at compile time
// k_ntt_pass / k_ntt_last above take the digit size at run time: which step is the first, which the last, whether a step is radix 2
// or 4, whether element k of it exists -- all of it is decided by branches around every element, and the compiler neither moves a load
// across a branch nor joins the loads of different elements: the first step waited for global memory once per element and table
// (eight dependent round trips to HBM per thread on a coset transform), the last step once per inter-pass twiddle.  The same passes
// with LOG_NP as a template parameter are straight-line code per step: all operands of a step (the 2^R elements, their coset shifts,
// the twiddles of the butterflies, the inter-pass twiddles of the outputs) are requested before the first of them is used.
// Same arithmetic in the same order: results are bit-identical to the generic kernels (which remain for every other shape; ZK_NTT_FIXED=0
// selects them everywhere).
template <int LOG_NP, int S, bool HAS_PRE>
__device__ __forceinline__ void ntt_pass_steps(const Lds29& L, const Fr* __restrict__ src, Fr* __restrict__ dst, const Tw29* __restrict__ tw,
                                               const Fr* __restrict__ pre, const Fr* __restrict__ out_tw, const int log_t, const uint64_t m, const uint64_t base) {
    constexpr int R = ((LOG_NP - S) & 1) ? 1 : 2, NE = 1 << R, hgt = 1 << S;
    constexpr bool FIRST = S == 0, LAST = S + R == LOG_NP;
    const int T = 1 << log_t, items = (1 << (LOG_NP + log_t)) >> R;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int c = it & (T - 1), b = it >> log_t;      // c fastest: T-element contiguous runs in global memory
        const int j = b & (hgt - 1);
        const int lo_d = ((b >> S) << (S + R)) | j;
        Fr otw[NE];
        if (LAST) {
#pragma unroll
            for (int k = 0; k < NE; ++k) otw[k] = ldg(out_tw + base + (uint64_t)(lo_d + k * hgt) * m + c);
        }
        Fr29 e[4];
        if (FIRST) {
            Fr raw[NE], praw[NE];
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const uint64_t gi = base + (uint64_t)bitrev(lo_d + k * hgt, LOG_NP) * m + c;
                raw[k] = ldg(src + gi);
                if (HAS_PRE) praw[k] = ldg(pre + gi);
            }
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                e[k] = unpack29<Fr29P>(raw[k]);
                if (HAS_PRE) e[k] = mul29(e[k], unpack29<Fr29P>(praw[k]));     // coset shift a[i] * g^i (table in R' form)
            }
            dit_first_step<R>(e, tw, LOG_NP);
        } else {
#pragma unroll
            for (int k = 0; k < NE; ++k) e[k] = L.load(((lo_d + k * hgt) << log_t) | c);
            dit_step<R>(e, tw, LOG_NP, S, j);
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int dl = lo_d + k * hgt;
            if (LAST) stg(dst + base + (uint64_t)dl * m + c, pack29_raw(mul29(e[k], unpack29<Fr29P>(otw[k]))));      // an intermediate (the next pass reads it): any representative below 2p will do, no conditional subtraction
            else L.store((dl << log_t) | c, e[k]);
        }
    }
    if constexpr (!LAST) {
        __syncthreads();
        ntt_pass_steps<LOG_NP, S + R, HAS_PRE>(L, src, dst, tw, pre, out_tw, log_t, m, base);
    }
}
template <int LOG_NP, bool HAS_PRE>
__global__ void __launch_bounds__(NTT_THREADS)
k_ntt_pass_f(NttIo io, const Tw29* __restrict__ tw, int log_t, int log_m, const Fr* __restrict__ pre, const Fr* __restrict__ out_tw, uint32_t ncols, int xcd_cols, int log_grp) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // (tile, column) numbering as in k_ntt_pass, with one more level: a tile of T < 4 columns reads and writes runs shorter than a
    // 128-byte line, and the 2^log_grp = 4 / T tiles that share those lines are consecutive workgroups of ONE XCD as well (same L2:
    // the line is fetched once and its parts are written back together), ahead of the columns of the launch
    uint32_t tile_id, col;
    if (xcd_cols) {
        const uint32_t slot = blockIdx.x >> 3, sub = slot & ((1u << log_grp) - 1u), s2 = slot >> log_grp;
        col = s2 % ncols;
        tile_id = ((((s2 / ncols) << 3) + (blockIdx.x & 7u)) << log_grp) + sub;
    } else { col = blockIdx.x % ncols; tile_id = blockIdx.x / ncols; }
    Lds29 L{smem, 1 << (LOG_NP + log_t)};
    const uint64_t m = 1ull << log_m;
    const uint32_t tiles_per_hi = (uint32_t)(m >> log_t);
    const uint32_t hi_idx = tile_id / tiles_per_hi, blk = tile_id % tiles_per_hi;
    const uint64_t base = ((uint64_t)hi_idx << (LOG_NP + log_m)) + ((uint64_t)blk << log_t);
    ntt_pass_steps<LOG_NP, 0, HAS_PRE>(L, io.src[col], io.dst[col], tw, pre, out_tw, log_t, m, base);
}

template <int LOG_NP, int S>
__device__ __forceinline__ void ntt_last_steps(const Lds29& L, const Fr* __restrict__ src, Fr* __restrict__ dst, const Tw29* __restrict__ tw,
                                               const int log_t, const int row, const uint32_t blk, const uint32_t mid, const int log_mid, const int log_n1) {
    constexpr int R = ((LOG_NP - S) & 1) ? 1 : 2, NE = 1 << R, hgt = 1 << S;
    constexpr bool FIRST = S == 0, LAST = S + R == LOG_NP;
    const int T = 1 << log_t, items = (1 << (LOG_NP + log_t)) >> R;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        int c, b;
        if (LAST) { c = it & (T - 1); b = it >> log_t; }                              // c fastest: coalesced output
        else { b = it & ((1 << (LOG_NP - R)) - 1); c = it >> (LOG_NP - R); }          // d fastest: conflict-free LDS
        const int j = b & (hgt - 1);
        const int lo_d = ((b >> S) << (S + R)) | j;
        const uint64_t i1 = ((uint64_t)blk << log_t) + c;
        Fr29 e[4];
        if (FIRST) {
            Fr raw[NE];
#pragma unroll
            for (int k = 0; k < NE; ++k) raw[k] = ldg(src + ((((i1 << log_mid) + mid) << LOG_NP) + bitrev(lo_d + k * hgt, LOG_NP)));
#pragma unroll
            for (int k = 0; k < NE; ++k) e[k] = unpack29<Fr29P>(raw[k]);
            dit_first_step<R>(e, tw, LOG_NP);
        } else {
#pragma unroll
            for (int k = 0; k < NE; ++k) e[k] = L.load(c * row + lo_d + k * hgt);
            dit_step<R>(e, tw, LOG_NP, S, j);
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int dl = lo_d + k * hgt;
            if (LAST) stg(dst + (i1 + (((uint64_t)mid + ((uint64_t)dl << log_mid)) << log_n1)), reduce_lazy29(e[k]));      // the output scale rode on the last inter-pass twiddle
            else L.store(c * row + dl, e[k]);
        }
    }
    if constexpr (!LAST) {
        __syncthreads();
        ntt_last_steps<LOG_NP, S + R>(L, src, dst, tw, log_t, row, blk, mid, log_mid, log_n1);
    }
}
template <int LOG_NP>
__global__ void __launch_bounds__(NTT_THREADS)
k_ntt_last_f(NttIo io, const Tw29* __restrict__ tw, int log_t, int log_n1, int log_mid, int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int row = (1 << LOG_NP) + ntt_row_pad(LOG_NP);
    Lds29 L{smem, (1 << log_t) * row};
    uint32_t bx = blockIdx.x;
    if (xcd_remap) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    const uint32_t mid = bx & ((1u << log_mid) - 1), blk = bx >> log_mid;
    ntt_last_steps<LOG_NP, 0>(L, io.src[blockIdx.y], io.dst[blockIdx.y], tw, log_t, row, blk, mid, log_mid, log_n1);
}

__global__ void k_scale(Fr* a, Fr s, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stg(a + i, ldg(a + i) * s);
}
int fr_scale_run(zk_ctx* ctx, Fr* d_a, const Fr& s, uint64_t n) {
    if (!n) return ZK_OK;
    hipLaunchKernelGGL(k_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_a, s, n);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}

// a[i] *= g^i   (EvaluationDomain::distribute_powers_zeta generalised), two-level table of g
__global__ void k_distribute_powers(const Fr* src, Fr* dst, const Fr* __restrict__ lo, const Fr* __restrict__ hi, int h, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    stg(dst + i, ldg(src + i) * two_level(lo, hi, h, (uint32_t)i));
}

// ----------------------------------------------------------------------------------- host side
static int build_powers(zk_ctx* ctx, const Fr& base, const Fr& mul, Fr* d_out, uint32_t count, int rprime) {
    hipLaunchKernelGGL(k_powers, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, base, mul, d_out, count, rprime);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}

static uint64_t domain_key(uint32_t log_n, const Fr& omega, const Fr* scale) {
    uint64_t hsh = 1469598103934665603ull;
    auto mix = [&](uint32_t v) { hsh = (hsh ^ v) * 1099511628211ull; };
    mix(log_n);
    for (int i = 0; i < 8; ++i) mix(omega.l[i]);
    if (scale) for (int i = 0; i < 8; ++i) mix(scale->l[i] ^ 0x9e3779b9u);
    return hsh;
}

static int get_domain(zk_ctx* ctx, uint32_t log_n, const Fr& omega, const Fr* scale, std::shared_ptr<NttDomain>* out) {
    const uint64_t key = domain_key(log_n, omega, scale);
    auto it = ctx->domains.find(key);
    if (it != ctx->domains.end()) { *out = it->second; return ZK_OK; }
    auto d = std::make_shared<NttDomain>();
    d->log_n = log_n;
    d->final_mul = to_rprime(scale ? *scale : Fr::one());
    const int P = log_n <= NTT_MAX_DIGIT ? 1 : (log_n <= 2 * NTT_MAX_DIGIT ? 2 : 3);
    d->npass = P;
    int rem = (int)log_n;
    for (int p = 0; p < P; ++p) {
        int b = (rem + (P - p) - 1) / (P - p);   // ceil split, big digits first
        d->pass[p].log_np = b;
        rem -= b;
        d->pass[p].log_m = rem;
    }
    // two-level tables (only needed when P > 1)
    if (P > 1) {
        d->h = (int)(log_n + 1) / 2;
        const uint32_t nlo = 1u << d->h, nhi = 1u << (log_n - d->h);
        ZK_HIP(ctx, hipMalloc(&d->d_lo, sizeof(Fr) * nlo));
        ZK_HIP(ctx, hipMalloc(&d->d_hi, sizeof(Fr) * nhi));
        int rc = build_powers(ctx, omega, Fr::one(), d->d_lo, nlo, 1);
        if (rc) return rc;
        Fr step = omega;
        for (int i = 0; i < d->h; ++i) step = sqr(step);
        rc = build_powers(ctx, step, Fr::one(), d->d_hi, nhi, 1);
        if (rc) return rc;
    }
    for (int p = 0; p < P; ++p) {
        const int b = d->pass[p].log_np;
        const uint32_t cnt = b ? (1u << (b - 1)) : 1u;
        ZK_HIP(ctx, hipMalloc(&d->d_tw[p], sizeof(Fr) * cnt));
        Fr w = omega;
        for (uint32_t i = 0; i < log_n - (uint32_t)b; ++i) w = sqr(w);   // omega^(n/n_p)
        int rc = build_powers(ctx, w, Fr::one(), d->d_tw[p], cnt, 1);
        if (rc) return rc;
        ZK_HIP(ctx, hipMalloc((void**)&d->d_tw29[p], sizeof(Tw29) * cnt));
        hipLaunchKernelGGL(k_split_twiddles, dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, (const Fr*)d->d_tw[p], d->d_tw29[p], cnt);
        ZK_CHECK_LAUNCH(ctx);
        d->pass[p].tw = d->d_tw29[p];
    }
    // full inter-pass twiddle tables (n x 32 B per non-last pass) while the domain is not huge
    if (P > 1 && log_n <= 24) {
        const uint64_t n = 1ull << log_n;
        for (int p = 0; p + 1 < P; ++p) {
            if (hipMalloc(&d->d_out_tw[p], sizeof(Fr) * n) != hipSuccess) { (void)hipGetLastError(); d->d_out_tw[p] = nullptr; break; }
            const NttPass& ps = d->pass[p];
            hipLaunchKernelGGL(k_build_out_twiddles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d->d_out_tw[p], (const Fr*)d->d_lo, (const Fr*)d->d_hi, d->h,
                               ps.log_np, ps.log_m, (int)log_n - ps.log_np - ps.log_m, n, d->final_mul, p == P - 2 ? 1 : 0);
            ZK_CHECK_LAUNCH(ctx);
            d->pass[p].out_tw = d->d_out_tw[p];
            if (p == P - 2) d->fin_folded = true;
        }
    }
    if (ctx->domains.size() > 64) ctx->domains.clear();
    ctx->domains[key] = d;
    *out = d;
    return ZK_OK;
}

static int set_lds_attr(zk_ctx* ctx) {
    if (ctx->ntt_attr_set) return ZK_OK;
    ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_pass, hipFuncAttributeMaxDynamicSharedMemorySize, NTT_TILE * NTT_LDS_BYTES_PER_ELT));
    ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_last, hipFuncAttributeMaxDynamicSharedMemorySize, (NTT_TILE + 128) * NTT_LDS_BYTES_PER_ELT));
    ctx->ntt_attr_set = true;
    return ZK_OK;
}


// launchers of the fixed-structure passes: false = this digit size has no instance (the caller takes the generic kernel)
template <int LOG_NP, bool HAS_PRE>
static bool launch_pass_f(zk_ctx* ctx, unsigned grid, unsigned threads, size_t lds, const NttIo& io, const Tw29* tw, int log_t, int log_m, const Fr* pre, const Fr* out_tw, uint32_t ncols, int xcd_cols, int log_grp) {
    const uint32_t bit = 1u << (2 * (LOG_NP - 7) + (HAS_PRE ? 1 : 0));          // the attribute is per device: remembered per context
    if (!(ctx->ntt_fixed_attr & bit)) {
        // a device that refuses the LDS size (not a gfx950, a lowered limit): this instance stays unused, the caller takes the generic kernel
        if (hipFuncSetAttribute((const void*)k_ntt_pass_f<LOG_NP, HAS_PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, NTT_TILE * NTT_LDS_BYTES_PER_ELT) != hipSuccess) { (void)hipGetLastError(); return false; }
        ctx->ntt_fixed_attr |= bit;
    }
    hipLaunchKernelGGL((k_ntt_pass_f<LOG_NP, HAS_PRE>), dim3(grid), dim3(threads), lds, ctx->stream, io, tw, log_t, log_m, pre, out_tw, ncols, xcd_cols, log_grp);
    return true;
}
static bool ntt_fixed_on() { const char* e = getenv("ZK_NTT_FIXED"); return !(e && atoi(e) == 0); }      // measurement knob, read per call (tests flip it inside one process)
static bool launch_pass_fixed(zk_ctx* ctx, int log_np, unsigned grid, unsigned threads, size_t lds, const NttIo& io, const Tw29* tw, int log_t, int log_m, const Fr* pre, const Fr* out_tw, uint32_t ncols, int xcd_cols, int log_grp) {
    if (!ntt_fixed_on() || !out_tw) return false;
#define ZK_PASS_CASE(N) case N: return pre ? launch_pass_f<N, true>(ctx, grid, threads, lds, io, tw, log_t, log_m, pre, out_tw, ncols, xcd_cols, log_grp) \
                                           : launch_pass_f<N, false>(ctx, grid, threads, lds, io, tw, log_t, log_m, pre, out_tw, ncols, xcd_cols, log_grp);
    switch (log_np) { ZK_PASS_CASE(7) ZK_PASS_CASE(8) ZK_PASS_CASE(9) ZK_PASS_CASE(10) ZK_PASS_CASE(11) default: return false; }      // 7, 8: the three-pass sizes (2^21 .. 2^24)
#undef ZK_PASS_CASE
}
template <int LOG_NP>
static bool launch_last_f(zk_ctx* ctx, dim3 grid, unsigned threads, size_t lds, const NttIo& io, const Tw29* tw, int log_t, int log_n1, int log_mid, int xcd_remap) {
    const uint32_t bit = 1u << (24 + LOG_NP - 7);
    if (!(ctx->ntt_fixed_attr & bit)) {
        if (hipFuncSetAttribute((const void*)k_ntt_last_f<LOG_NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (NTT_TILE + 128) * NTT_LDS_BYTES_PER_ELT) != hipSuccess) { (void)hipGetLastError(); return false; }
        ctx->ntt_fixed_attr |= bit;
    }
    hipLaunchKernelGGL((k_ntt_last_f<LOG_NP>), grid, dim3(threads), lds, ctx->stream, io, tw, log_t, log_n1, log_mid, xcd_remap);
    return true;
}
static bool launch_last_fixed(zk_ctx* ctx, int log_np, dim3 grid, unsigned threads, size_t lds, const NttIo& io, const Tw29* tw, int log_t, int log_n1, int log_mid, int xcd_remap) {
    if (!ntt_fixed_on()) return false;
#define ZK_LAST_CASE(N) case N: return launch_last_f<N>(ctx, grid, threads, lds, io, tw, log_t, log_n1, log_mid, xcd_remap);
    switch (log_np) { ZK_LAST_CASE(7) ZK_LAST_CASE(8) ZK_LAST_CASE(9) ZK_LAST_CASE(10) ZK_LAST_CASE(11) default: return false; }
#undef ZK_LAST_CASE
}

static int pick_threads(int tile) { return tile >= 4096 ? 1024 : (tile >= 1024 ? 512 : (tile >= 256 ? 128 : 64)); }

// Generic driver.  `scale` (nullable) multiplies every output; coset_pre (nullable): a[i] *= g^i
// before the transform; coset_post (nullable): out[i] *= g^i after it.  d_src (nullable): the input
// is read from d_src and d_data only receives the result (out of place, no extra copy).
// `count` transforms over the same domain, in place (d_src == nullptr) or from d_src[i] to d_data[i]; launched
// NTT_BATCH columns at a time.  coset_post and the unfused coset shift are per-column passes and only taken by the
// single-column entry point.
int ntt_run_many(zk_ctx* ctx, Fr* const* d_datas, const Fr* const* d_srcs, size_t count, uint32_t log_n, const Fr& omega, const Fr* scale, const Fr* coset_pre, const Fr* coset_post, bool fuse_pre);
int ntt_run(zk_ctx* ctx, Fr* d_data, uint32_t log_n, const Fr& omega, const Fr* scale, const Fr* coset_pre, const Fr* coset_post, const Fr* d_src, bool fuse_pre) {
    return ntt_run_many(ctx, &d_data, d_src ? &d_src : nullptr, 1, log_n, omega, scale, coset_pre, coset_post, fuse_pre);
}
int ntt_run_many(zk_ctx* ctx, Fr* const* d_datas, const Fr* const* d_srcs, size_t count, uint32_t log_n, const Fr& omega, const Fr* scale, const Fr* coset_pre, const Fr* coset_post, bool fuse_pre) {
    if (count == 0) return ZK_OK;
    if (count > 1 && (log_n == 0 || coset_post)) {          // rare shapes: one by one
        for (size_t i = 0; i < count; ++i) {
            const Fr* one_src = d_srcs ? d_srcs[i] : nullptr;
            int r = ntt_run_many(ctx, d_datas + i, one_src ? &one_src : nullptr, 1, log_n, omega, scale, coset_pre, coset_post, fuse_pre);
            if (r) return r;
        }
        return ZK_OK;
    }
    Fr* d_data = d_datas[0];
    const Fr* d_src = d_srcs ? d_srcs[0] : nullptr;
    if (log_n == 0) {   // size-1 transform: identity (times the scale)
        if (d_src && d_src != d_data) ZK_HIP(ctx, hipMemcpyAsync(d_data, d_src, sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
        if (scale) {
            hipLaunchKernelGGL(k_scale, dim3(1), dim3(64), 0, ctx->stream, d_data, *scale, (uint64_t)1);
            ZK_CHECK_LAUNCH(ctx);
        }
        return ZK_OK;
    }
    int rc = set_lds_attr(ctx);
    if (rc) return rc;
    const uint64_t n = 1ull << log_n;

    auto run_distribute = [&](const Fr& g, const Fr* from) -> int {
        const int h = (int)(log_n + 1) / 2;
        const uint32_t nlo = 1u << h, nhi = 1u << (log_n - h);
        // the coset generators are a handful of constants (zeta, zeta^-1): keep their tables
        const uint64_t key = domain_key(log_n, g, nullptr) ^ 0xC05E7ull;
        Fr* tab = nullptr;
        auto it = ctx->pow_tables.find(key);
        if (it != ctx->pow_tables.end()) {
            tab = (Fr*)it->second;
        } else {
            if (hipMalloc(&tab, sizeof(Fr) * ((size_t)nlo + nhi)) != hipSuccess) return ctx->fail(ZK_ERR_OOM, "coset table allocation failed");
            int r = build_powers(ctx, g, Fr::one(), tab, nlo, 0);
            if (r) return r;
            Fr step = g;
            for (int i = 0; i < h; ++i) step = sqr(step);
            r = build_powers(ctx, step, Fr::one(), tab + nlo, nhi, 0);
            if (r) return r;
            ctx->pow_tables[key] = tab;
        }
        hipLaunchKernelGGL(k_distribute_powers, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, from, d_data, tab, tab + nlo, h, n);
        ZK_CHECK_LAUNCH(ctx);
        return ZK_OK;
    };
    const Fr* cur = d_src ? d_src : d_data;
    const Fr* pre_table = nullptr;       // full table g^i (R' form), multiplied in by the first pass as it loads
    if (coset_pre && fuse_pre) {
        const uint64_t key = domain_key(log_n, *coset_pre, nullptr) ^ 0xF0117ABull;
        auto it = ctx->pow_tables.find(key);
        if (it != ctx->pow_tables.end()) {
            pre_table = (const Fr*)it->second;
        } else {
            Fr* tab = nullptr;
            if (hipMalloc(&tab, sizeof(Fr) * n) != hipSuccess) { (void)hipGetLastError(); tab = nullptr; }    // no memory: separate pass below
            if (tab) {
                int r = build_powers(ctx, *coset_pre, Fr::one(), tab, (uint32_t)n, 1);
                if (r) return r;
                ctx->pow_tables[key] = tab;
                pre_table = tab;
            }
        }
    }
    if (coset_pre && !pre_table) {
        if (count > 1) {                                     // no memory for the shift table: the columns go one by one through the separate pass
            for (size_t i = 0; i < count; ++i) {
                const Fr* one_src = d_srcs ? d_srcs[i] : nullptr;
                int r = ntt_run_many(ctx, d_datas + i, one_src ? &one_src : nullptr, 1, log_n, omega, scale, coset_pre, coset_post, fuse_pre);
                if (r) return r;
            }
            return ZK_OK;
        }
        rc = run_distribute(*coset_pre, cur); if (rc) return rc; cur = d_data;
    }

    std::shared_ptr<NttDomain> dom;
    rc = get_domain(ctx, log_n, omega, scale, &dom);
    if (rc) return rc;
    const int P = dom->npass;
    // columns per launch: enough tiles to give every CU a few workgroups, bounded by the scratch it takes
    size_t per_launch = 1;
    if (count > 1) {
        const uint64_t tiles = std::max<uint64_t>(1, n >> 12);
        per_launch = (size_t)std::min<uint64_t>(NTT_BATCH, std::max<uint64_t>(1, 4096 / tiles));       // sixteen columns at 2^20: a launch's last round of workgroups and the 10-20 us between dependent launches are paid once per sixteen columns (headline proof, alternating A/B on one box: 1.006 s with four, 0.9855 with eight, 0.9793 with sixteen; alone the transform does not care: 94.8 / 94.6 us)
        if (const char* e = getenv("ZK_NTT_BATCH")) { const int v = atoi(e); if (v >= 1 && v <= NTT_BATCH) per_launch = (size_t)v; }      // measurement knob
    }
    Fr* scratch = nullptr;
    if (P > 1) {
        // transforms on the auxiliary stream run BESIDE transforms of the main stream (coset transforms ahead of the quotient,
        // prover.hip): the intermediate buffer of a multi-pass transform is per stream
        // sixteen columns per launch take 512 MiB of scratch per stream at 2^20: on a device that cannot spare it, fewer columns per launch (slower, not fatal)
        const int slot = ctx->stream_aux && ctx->stream == ctx->stream_aux ? SC_NTT_AUX : SC_NTT;
        for (;;) {
            scratch = (Fr*)ctx->get_scratch(slot, sizeof(Fr) * n * per_launch);
            if (scratch || per_launch == 1) break;
            (void)hipGetLastError();
            per_launch = (per_launch + 1) / 2;
        }
        if (!scratch) return ZK_ERR_OOM;
    }
    for (size_t first = 0; first < count; first += per_launch) {
        const size_t nb = std::min(per_launch, count - first);
        NttIo io_in{}, io_mid{}, io_last{};
        // buffers: P=1: data->data (whole transform inside one workgroup, safe in place)
        //          P=2: data->scratch, scratch->data;   P=3: data->data, data->scratch, scratch->data
        for (size_t j = 0; j < nb; ++j) {
            Fr* dd = d_datas[first + j];
            const Fr* ss = (count == 1) ? cur : (d_srcs && d_srcs[first + j] ? d_srcs[first + j] : dd);
            io_in.src[j] = ss;
            io_in.dst[j] = dd;
        }
        NttIo cur_io = io_in;                                  // .src = where the next pass reads
        for (int p = 0; p + 1 < P; ++p) {
            const NttPass& ps = dom->pass[p];
            // tile of the strided passes: 4096 elements (T = 4: 128-byte runs) when there are two passes; with three passes
            // 2048 elements measured 8-10 % faster (two workgroups per CU: one loads while the other computes)
            int log_tile_pass = P == 3 ? 11 : 12;
            if (const char* e = getenv("ZK_NTT_PASS_LOGTILE")) { const int v = atoi(e); if (v >= 10 && v <= 12) log_tile_pass = v; }    // measurement knob
            int log_t = log_tile_pass - ps.log_np;
            if (log_t < 0) log_t = 0;
            if (log_t > ps.log_m) log_t = ps.log_m;
            const int tile = 1 << (ps.log_np + log_t);
            NttIo io{};
            for (size_t j = 0; j < nb; ++j) {
                io.src[j] = cur_io.src[j];
                io.dst[j] = (p == P - 2) ? scratch + j * n : d_datas[first + j];
            }
            const unsigned blocks = (unsigned)(n >> (ps.log_np + log_t));
            const int tw_shift = (int)log_n - ps.log_np - ps.log_m;
            ZkProfScope pscope(ctx, "ntt_pass");
            pscope.bytes = (uint64_t)nb * n * 64 / (uint64_t)P;      // a transform's algorithmic 64 B per element (read once, write once), spread over its P launches
            static const bool xcd_off = getenv("ZK_NTT_XCD_COLS") && atoi(getenv("ZK_NTT_XCD_COLS")) == 0;       // measurement knob
            const int xcd_cols = (nb > 1 && blocks % 8 == 0 && !xcd_off) ? 1 : 0;
            const int log_grp = log_t < 2 ? 2 - log_t : 0;           // tiles per 128-byte line of a run
            const int xcd_fixed = (!xcd_off && blocks % (8u << log_grp) == 0 && (nb > 1 || log_grp > 0)) ? 1 : 0;
            if (!launch_pass_fixed(ctx, ps.log_np, blocks * (unsigned)nb, (unsigned)std::max(64, std::min(1024, tile >> 2)) /* one radix-4 item per thread and step */, (size_t)tile * NTT_LDS_BYTES_PER_ELT, io, ps.tw, log_t, ps.log_m,
                                   p == 0 ? pre_table : (const Fr*)nullptr, ps.out_tw, (uint32_t)nb, xcd_fixed, xcd_fixed ? log_grp : 0))
                hipLaunchKernelGGL(k_ntt_pass, dim3(blocks * (unsigned)nb), dim3(pick_threads(tile)), (size_t)tile * NTT_LDS_BYTES_PER_ELT, ctx->stream, io, ps.tw,
                                   dom->d_lo, dom->d_hi, dom->h, ps.log_np, log_t, ps.log_m, tw_shift, p == 0 ? pre_table : (const Fr*)nullptr, ps.out_tw,
                                   (uint32_t)nb, xcd_cols);
            ZK_CHECK_LAUNCH(ctx);
            for (size_t j = 0; j < nb; ++j) cur_io.src[j] = io.dst[j];
        }
        {
            const NttPass& ps = dom->pass[P - 1];
            const int log_n1 = P == 1 ? 0 : dom->pass[0].log_np;
            const int log_mid = P == 3 ? dom->pass[1].log_np : 0;
            // the last pass reads whole rows: a 2048-element tile (two rows of 2^10, 72 KiB of LDS) lets two workgroups share a CU
            // and overlap their load / compute / store phases: -4 % at 2^20, -5 % at 2^22
            int log_tile_last = 11;
            if (const char* e = getenv("ZK_NTT_LAST_LOGTILE")) { const int v = atoi(e); if (v >= 10 && v <= 12) log_tile_last = v; }    // measurement knob
            int log_t = log_tile_last - ps.log_np;
            if (log_t < 0) log_t = 0;
            if (log_t > log_n1) log_t = log_n1;
            const int tile = 1 << (ps.log_np + log_t);
            const unsigned blocks = (unsigned)(n >> (ps.log_np + log_t));
            NttIo io{};
            for (size_t j = 0; j < nb; ++j) { io.src[j] = cur_io.src[j]; io.dst[j] = d_datas[first + j]; }
            ZkProfScope pscope(ctx, "ntt_last");
            pscope.bytes = (uint64_t)nb * n * 64 / (uint64_t)P;
            const int xcd_last = (log_mid == 0 && blocks % 8 == 0 && blocks >= 16 && ntt_xcd_remap()) ? 1 : 0;
            const size_t lds_last = (size_t)(tile + (ntt_row_pad(ps.log_np) << log_t)) * NTT_LDS_BYTES_PER_ELT;
            if (!(P > 1 && dom->fin_folded && launch_last_fixed(ctx, ps.log_np, dim3(blocks, (unsigned)nb), (unsigned)std::max(64, std::min(1024, tile >> 2)), lds_last, io, ps.tw, log_t, log_n1, log_mid, xcd_last)))
                hipLaunchKernelGGL(k_ntt_last, dim3(blocks, (unsigned)nb), dim3(pick_threads(tile)), lds_last, ctx->stream, io, ps.tw,
                                   ps.log_np, log_t, log_n1, log_mid, dom->final_mul, P == 1 ? pre_table : (const Fr*)nullptr, dom->fin_folded ? 1 : 0, xcd_last);
            ZK_CHECK_LAUNCH(ctx);
        }
    }
    if (coset_post) { rc = run_distribute(*coset_post, d_data); if (rc) return rc; }
    return ZK_OK;
}

// ---- one transform spread over several GPUs (SURVEY 8e "domain halves": the 4-step split) ------
// n = W * m with W = world.  Write i = i1 + W * i2 and j = j2 + m * j1:
//   X[j2 + m j1] = sum_{i1} w_W^(i1 j1) * [ w_n^(i1 j2) * sum_{i2} x[i1 + W i2] w_m^(i2 j2) ]
// Rank i1 owns the residue class x[i1 + W i2]: a local size-m transform with the twiddle
// w_n^(i1 j2) folded in as its output scaling, ONE all-to-all (rank c receives the j2 slice
// [c m/W, (c+1) m/W) from everyone), and W-point butterflies across the received rows.
template <int W>
__global__ void __launch_bounds__(256) k_ntt_cross(const Fr* __restrict__ recv, Fr* __restrict__ out, size_t cols, const Fr* __restrict__ tw /* w_W^t, t < W/2 */, Fr scale, int scaled) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    constexpr int LOGW = W == 2 ? 1 : W == 4 ? 2 : W == 8 ? 3 : 4;
    Fr a[W];
#pragma unroll
    for (int i = 0; i < W; ++i) {
        int rev = 0;
#pragma unroll
        for (int b = 0; b < LOGW; ++b) rev |= ((i >> b) & 1) << (LOGW - 1 - b);
        a[rev] = ldg(recv + (size_t)i * cols + c);
    }
#pragma unroll
    for (int st = 1; st <= LOGW; ++st) {
#pragma unroll
        for (int h = 0; h < W / 2; ++h) {                  // butterfly h of this stage: block h / half, lane t
            const int half = 1 << (st - 1), t = h & (half - 1), lo = ((h >> (st - 1)) << st) + t, hi = lo + half;
            const Fr u = a[lo];
            const Fr v = t == 0 ? a[hi] : a[hi] * tw[t << (LOGW - st)];
            a[lo] = u + v;
            a[hi] = u - v;
        }
    }
#pragma unroll
    for (int j = 0; j < W; ++j) stg(out + (size_t)j * cols + c, scaled ? a[j] * scale : a[j]);
}

}  // namespace zk

using namespace zk;
extern "C" int zk_ntt_sharded(zk_ctx* ctx, void* d_local, uint32_t log_n, int inverse, uint32_t rank, uint32_t world, zk_alltoall_fn exchange, void* user) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_local, "null pointer");
    ZK_REQUIRE(ctx, exchange || world == 1 || (comm_ready(ctx) && ctx->comm_world == world && ctx->comm_rank == rank), "no all-to-all callback and no matching communicator (zk_comm_init)");
    ZK_REQUIRE(ctx, world >= 1 && world <= 16 && (world & (world - 1)) == 0 && rank < world, "world must be a power of two <= 16, rank < world");
    ZK_REQUIRE(ctx, log_n <= 28, "log_n exceeds the two-adicity of Fr (28)");
    uint32_t log_w = 0;
    while ((1u << log_w) < world) ++log_w;
    ZK_REQUIRE(ctx, log_n >= 2 * log_w, "need n >= world^2");
    if (world == 1) return zk_ntt(ctx, d_local, log_n, inverse);
    const uint32_t log_m = log_n - log_w;
    const size_t m = (size_t)1 << log_m, cols = m >> log_w;
    Fr w_n = fr_root_of_unity(log_n), w_m = fr_root_of_unity(log_m), w_w = fr_root_of_unity(log_w);
    if (inverse) { w_n = fr_inv_host(w_n); w_m = fr_inv_host(w_m); w_w = fr_inv_host(w_w); }
    // steps 1 + 2: local transform over <w_m>, then element j2 times w_n^(rank * j2)
    const Fr g = fr_pow(w_n, rank);
    int rc = ntt_run(ctx, (Fr*)d_local, log_m, w_m, nullptr, nullptr, rank ? &g : nullptr);
    if (rc) return rc;
    // step 3: the exchange.  d_local is already [peer][cols]; the callback returns with d_recv complete
    Fr* recv = (Fr*)ctx->pool_get(m * sizeof(Fr));
    Fr* d_tw = (Fr*)ctx->pool_get(16 * sizeof(Fr));
    if (!recv || !d_tw) { ctx->pool_put(recv, m * sizeof(Fr)); ctx->pool_put(d_tw, 16 * sizeof(Fr)); return ctx->fail(ZK_ERR_OOM, "sharded NTT: exchange buffer of %zu bytes", m * sizeof(Fr)); }
    auto done = [&](int code) { ctx->pool_put(recv, m * sizeof(Fr)); ctx->pool_put(d_tw, 16 * sizeof(Fr)); return code; };
    hipError_t e = hipSuccess;
    if (exchange) {
        e = hipStreamSynchronize(ctx->stream);       // a callback runs on the caller's own stream
        if (e != hipSuccess) return done(ctx->fail(ZK_ERR_HIP, "sharded NTT: %s", hipGetErrorString(e)));
        if (exchange(user, d_local, cols * sizeof(Fr), recv) != 0) return done(ctx->fail(ZK_ERR_INVALID_ARG, "sharded NTT: the all-to-all callback failed"));
    } else {
        // in-library RCCL: W - 1 grouped send / receive pairs on this stream, no host synchronisation between
        // the local transform, the exchange and the cross butterflies
        rc = comm_alltoall_dev(ctx, d_local, cols * sizeof(Fr), recv);
        if (rc) return done(rc);
    }
    // step 4: W-point transforms across the received rows, 1/n folded in for the inverse
    Fr tw[8];
    tw[0] = Fr::one();
    for (uint32_t t = 1; t < (world > 1 ? world / 2 : 1); ++t) tw[t] = tw[t - 1] * w_w;
    e = hipMemcpyAsync(d_tw, tw, sizeof(Fr) * (world / 2 ? world / 2 : 1), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return done(ctx->fail(ZK_ERR_HIP, "sharded NTT: %s", hipGetErrorString(e)));
    const Fr ninv = inverse ? fr_inv_host(fr_from_u64(1ull << log_n)) : Fr::one();
    const dim3 grid((unsigned)((cols + 255) / 256)), block(256);
    switch (world) {
        case 2: hipLaunchKernelGGL(k_ntt_cross<2>, grid, block, 0, ctx->stream, (const Fr*)recv, (Fr*)d_local, cols, (const Fr*)d_tw, ninv, inverse); break;
        case 4: hipLaunchKernelGGL(k_ntt_cross<4>, grid, block, 0, ctx->stream, (const Fr*)recv, (Fr*)d_local, cols, (const Fr*)d_tw, ninv, inverse); break;
        case 8: hipLaunchKernelGGL(k_ntt_cross<8>, grid, block, 0, ctx->stream, (const Fr*)recv, (Fr*)d_local, cols, (const Fr*)d_tw, ninv, inverse); break;
        default: hipLaunchKernelGGL(k_ntt_cross<16>, grid, block, 0, ctx->stream, (const Fr*)recv, (Fr*)d_local, cols, (const Fr*)d_tw, ninv, inverse); break;
    }
    e = hipGetLastError();
    if (e != hipSuccess) return done(ctx->fail(ZK_ERR_HIP, "sharded NTT: %s", hipGetErrorString(e)));
    e = hipStreamSynchronize(ctx->stream);     // tw lives on this frame
    if (e != hipSuccess) return done(ctx->fail(ZK_ERR_HIP, "sharded NTT: %s", hipGetErrorString(e)));
    return done(ZK_OK);
}
