// BN254 G1 multi-scalar multiplication for gfx950: halo2_proofs::arithmetic::best_multiexp
// (external crate; reached from ParamsKZG::commit / commit_lagrange, reference
// call sites A1-A4).  result = sum_i scalars[i] * bases[i].
//
// Pippenger, MI355X layout:
//   1. digits    scalars leave Montgomery form (one Montgomery product by 1) and are recoded into
//                W = ceil(256/c) signed c-bit digits, stored as a window-major u16 matrix; every
//                non-zero digit is a (bucket, point) pair, bucket = window * 2^(c-1) + |d| - 1.
//   2. sort      counting sort by bucket with LDS-privatised counters: workgroup (range, window,
//                slice) streams its quarter of the digit row twice (count, scatter); no global
//                atomics; XCD-aware placement; empty windows are skipped.
//   3. buckets   buckets are ordered by size and split into tasks of <= 48 points (skew-proof);
//                one lane per task gathers R'-form affine bases (64 B each) and accumulates with
//                mixed XYZZ additions on 29-bit limbs (8M + 2S, no inversions); the partials of
//                multi-task buckets are combined per wave (segmented shuffle reduction), then
//                per bucket.
//   4. reduce    per window: sum_b (b+1) * B_b by running sums + LDS tree.  For SRS bases the
//                fixed-base window tables (2^(c*w) * P_i precomputed) make every window's bucket b
//                carry the same weight, so the W bucket arrays are folded first and ONE window is
//                reduced.  Runs on a side stream: in a batch it hides under the next MSM.
//   M. merged    with window tables (SRS bases) all windows share ONE bucket set: every non-zero
//                digit of every window is an entry (bucket, table index w * 2^k + i); the window
//                size is then free of the bucket-count / window-count trade-off (c = 20 at 2^20:
//                13 windows instead of 16, 2^19 buckets of ~26 entries), nothing is folded, and
//                the entries are sorted by a two-level counting sort (partition by the high
//                bucket bits while the digits are produced, then an LDS counting sort per
//                partition) that touches every entry a constant number of times.
//   5. tail      window sums go to the host: Horner over the windows (c doublings each; a
//                dependent doubling chain is issue-bound on one GPU lane) and the affine
//                normalisation; with window tables only the normalisation is left.
#include <chrono>
#include <cmath>

#include "ctx.hpp"
#include "ec29.hip.hpp"
#include "host_fq.hpp"

namespace zk {

constexpr int MSM_MAX_C = 16;
constexpr uint32_t NEG_BIT = 0x80000000u;

struct MsmPlan {
    int c;          // window bits
    int W;          // windows
    uint32_t B;     // buckets per window = 2^(c-1)
    int top_shift = 0;   // merged-window path: the top window's digit is scaled by 2^top_shift (its table entry is 2^(c (W-1) - top_shift) P)
};

// Graph replay of small batches (msm_batch_merged, graph mode): what differs from column to column -- where the scalars are,
// where the result goes -- is read from a device-side table, cols[*ctr], so that the launch sequence of a column can be
// captured once and replayed with one API call; the last kernel of the sequence advances *ctr to the pipeline's next column.
struct MsmCol { const Fr* scalars; G1Xyzz* out; };

static MsmPlan make_plan(size_t n) {
    int lg = 0;
    while ((1ull << (lg + 1)) <= n) ++lg;
    int c = lg - 4;
    if (c < 4) c = 4;
    if (c > MSM_MAX_C) c = MSM_MAX_C;
    MsmPlan p;
    p.c = c;
    p.W = (256 + c - 1) / c;
    p.B = 1u << (c - 1);
    return p;
}

// ---- sort phase --------------------------------------------------------------------------------
// 1. k_msm_digits: every scalar leaves Montgomery form once and is recoded; digit codes go to a
//    window-major u16 matrix dig[w][i] (row stride n_pad, multiple of 64):
//      0xFFFF = zero digit, otherwise bit 15 = sign, bits 0..14 = bucket (|d| - 1).
// 2. k_msm_lds_count / k_msm_lds_scatter: workgroup (r, w) owns bucket range r of window w
//    (<= 2048 buckets).  It streams the whole digit row (2 MiB at 2^20, L2-resident, 16 B/lane)
//    and keeps only digits in its range: counters / cursors live in LDS (ds_add_rtn_u32), so the
//    sort needs no global atomics, and a workgroup's slice of `idx` is written by that workgroup
//    alone -- partial lines merge in its XCD's L2 instead of costing one 64 B HBM write per index.
//    Every workgroup does the same n digit tests whatever the scalar distribution: no skew.
constexpr uint32_t DIG_ZERO = 0xFFFFu;
constexpr int MSM_RANGE_MAX_BITS = 11;   // <= 2048 buckets per workgroup

// wflag[w] is raised when window w holds at least one non-zero digit: the sweeps skip the others
// (selector / boolean / small-value columns leave most windows empty).
template <int C>
__device__ __forceinline__ void recode_all(const Fr& s, uint32_t (&code)[(256 + C - 1) / C]) {
    constexpr int W = (256 + C - 1) / C;      // window bits are a template parameter: every limb index below is static
    constexpr uint32_t mask = (1u << C) - 1, half = 1u << (C - 1);
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int bit = w * C, limb = bit >> 5, sh = bit & 31;
        uint32_t d = limb < 8 ? (s.l[limb < 8 ? limb : 7] >> sh) : 0u;
        if (sh + C > 32 && limb + 1 < 8) d |= s.l[limb + 1 < 8 ? limb + 1 : 7] << (32 - sh);
        d = (d & mask) + carry;
        if (d > half) { carry = 1; const uint32_t mag = (1u << C) - d; code[w] = mag ? (0x8000u | (mag - 1)) : DIG_ZERO; }
        else { carry = 0; code[w] = d ? (d - 1) : DIG_ZERO; }
    }
}
// One lane recodes TWO consecutive scalars and stores their digits as one 32-bit word per window:
// a wave writes 256 contiguous bytes per row with dword stores (2-byte stores, one scalar per
// lane, ran at a third of the rate).  n_pad is even.
template <int C>
__global__ void __launch_bounds__(256) k_msm_digits(const Fr* __restrict__ scalars_arg, uint64_t n, uint64_t n_pad, uint16_t* __restrict__ dig, uint32_t* __restrict__ wflag,
                                                    const MsmCol* __restrict__ cols = nullptr, const uint32_t* __restrict__ ctr = nullptr) {
    constexpr int W = (256 + C - 1) / C;
    const Fr* __restrict__ scalars = cols ? cols[*ctr].scalars : scalars_arg;
    const uint64_t i = 2 * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_pad) return;
    uint32_t c0[W], c1[W];
    if (i < n) recode_all<C>(from_mont(ldg(scalars + i)), c0);
    else {
#pragma unroll
        for (int w = 0; w < W; ++w) c0[w] = DIG_ZERO;
    }
    if (i + 1 < n) recode_all<C>(from_mont(ldg(scalars + i + 1)), c1);
    else {
#pragma unroll
        for (int w = 0; w < W; ++w) c1[w] = DIG_ZERO;
    }
    uint32_t* dig32 = reinterpret_cast<uint32_t*>(dig);
#pragma unroll
    for (int w = 0; w < W; ++w) {
        dig32[((uint64_t)w * n_pad + i) >> 1] = c0[w] | (c1[w] << 16);
        const uint64_t any = __ballot(c0[w] != DIG_ZERO || c1[w] != DIG_ZERO);
        // one lane per wave raises the flag, and only while it still reads 0
        if (any && (uint32_t)__builtin_ctzll(any) == (threadIdx.x & 63u) && wflag[w] == 0u) wflag[w] = 1u;
    }
}
static void launch_digits(int c, dim3 grid, hipStream_t st, const Fr* scalars, uint64_t n, uint64_t n_pad, uint16_t* dig, uint32_t* wflag, const MsmCol* cols = nullptr, const uint32_t* ctr = nullptr) {
#define ZK_DIG_CASE(C) case C: hipLaunchKernelGGL(k_msm_digits<C>, grid, dim3(256), 0, st, scalars, n, n_pad, dig, wflag, cols, ctr); break;
    switch (c) {
        ZK_DIG_CASE(4) ZK_DIG_CASE(5) ZK_DIG_CASE(6) ZK_DIG_CASE(7) ZK_DIG_CASE(8) ZK_DIG_CASE(9) ZK_DIG_CASE(10)
        ZK_DIG_CASE(11) ZK_DIG_CASE(12) ZK_DIG_CASE(13) ZK_DIG_CASE(14) ZK_DIG_CASE(15) ZK_DIG_CASE(16)
    }
#undef ZK_DIG_CASE
}

// A row is also cut into MSM_SLICES slices (workgroup = range x slice): witness columns leave most
// windows empty, and with one workgroup per (range, window) the few non-empty windows would keep
// only a quarter of the CUs busy while each still streams a whole row.  Counters are laid out
// [bucket][slice], so one exclusive scan over the flat array yields every (bucket, slice) cursor.
constexpr int MSM_SLICES = 4;        // == SCAN_ITEMS: a scan thread sees the slices of one bucket
template <bool SCATTER>
__global__ void __launch_bounds__(1024) k_msm_lds_sweep(const uint16_t* __restrict__ dig, uint64_t n_pad, int range_bits, uint32_t B,
                                                        uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets, uint32_t* __restrict__ idx, const uint32_t* __restrict__ wflag, uint32_t nwin) {
    __shared__ uint32_t lds[1 << MSM_RANGE_MAX_BITS];
    // XCD-aware placement: workgroups go to the 8 XCDs round-robin by linear id, so id = xcd + 8 * j
    // with window = xcd + 8 * (j / ranges): all range-workgroups of a window share one XCD and its
    // L2 serves the window's digit row to all of them (read from HBM / MALL once, not once per XCD).
    const uint32_t nranges = B >> range_bits, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t sl = slot % MSM_SLICES, r = (slot / MSM_SLICES) % nranges, w = xcd + 8u * (slot / (MSM_SLICES * nranges));
    const uint32_t range = 1u << range_bits, rmask = range - 1;
    if (w >= nwin) return;
    const uint64_t gbase = (uint64_t)w * B + ((uint64_t)r << range_bits);
    if (wflag[w] == 0u) {      // empty window: nothing to count or place
        if (!SCATTER) for (uint32_t t = threadIdx.x; t < range; t += blockDim.x) counts[(gbase + t) * MSM_SLICES + sl] = 0u;
        return;
    }
    for (uint32_t t = threadIdx.x; t < range; t += blockDim.x) lds[t] = SCATTER ? offsets[(gbase + t) * MSM_SLICES + sl] : 0u;
    __syncthreads();
    // 16 digits per lane and trip, two digits per 32-bit word.  A digit belongs to this workgroup
    // when the range field of its code (bits range_bits..14) equals r: xor with the wanted field,
    // mask, and a carry trick turn "half-word == 0" into one flag bit per digit (~3 VALU per digit
    // instead of a shift / mask / compare chain per digit).  The hits of a lane are then walked
    // with a loop that runs max-over-lanes(hits) times, so the LDS atomics go out with several
    // active lanes each.  The next trip's loads are issued before the current one is processed.
    const uint32_t field = (0x7FFFu & ~rmask) * 0x00010001u, want = (r << range_bits) * 0x00010001u;
    const bool zero_aliases = (0x7FFFu >> range_bits) == r;       // DIG_ZERO = 0xFFFF carries this range's field
    const uint4* row = reinterpret_cast<const uint4*>(dig + (uint64_t)w * n_pad);
    const uint64_t nvec_slice = n_pad >> 4 >> 2, nvec = (uint64_t)(sl + 1) * nvec_slice;   // n_pad is a multiple of 16 * MSM_SLICES
    static_assert(MSM_SLICES == 4, "slice arithmetic above assumes 4 slices");
    uint64_t v = (uint64_t)sl * nvec_slice + threadIdx.x;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    if (v < nvec) { q0 = row[2 * v]; q1 = row[2 * v + 1]; }
    while (v < nvec) {
        const uint32_t wd[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const uint64_t vn = v + blockDim.x;
        if (vn < nvec) { q0 = row[2 * vn]; q1 = row[2 * vn + 1]; }
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t z = (wd[j] ^ want) & field;                    // half == 0  <=>  digit in range
            uint32_t nz = (z + 0x7FFF7FFFu) & 0x80008000u;                 // bit 15 / 31  <=>  half != 0
            if (zero_aliases) {
                if ((wd[j] & 0xFFFFu) == DIG_ZERO) nz |= 0x8000u;
                if ((wd[j] >> 16) == DIG_ZERO) nz |= 0x80000000u;
            }
            m = (m >> 2) | nz;                                            // word j ends at bits 1 + 2j / 17 + 2j
        }
        uint32_t hits = ~m & 0xAAAAAAAAu;
        while (hits) {
            const uint32_t p = (uint32_t)__builtin_ctz(hits);
            hits &= hits - 1;
            const uint32_t j = (p & 15u) >> 1, hi = p >> 4;
            uint32_t word = wd[0];
#pragma unroll
            for (uint32_t t = 1; t < 8; ++t) word = j == t ? wd[t] : word;
            const uint32_t code = (word >> (hi * 16)) & 0xFFFFu;
            // Runs of equal scalars (grand products that stay constant over unused rows, selector
            // columns) put the same bucket in every lane, and 64 atomics on one LDS word serialise.
            // Peel the first active lane's bucket: its lanes share one atomic; the rest go alone.
            const uint32_t slot = code & rmask;
            const uint32_t lead = __builtin_amdgcn_readfirstlane(slot);
            const uint64_t same = __ballot(slot == lead);
            const uint32_t lane = threadIdx.x & 63u;
            uint32_t pos = 0;
            if (slot == lead) {
                const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                if (rank == 0) pos = atomicAdd(&lds[slot], (uint32_t)__popcll(same));
                if (SCATTER) pos = __shfl(pos, (int)__builtin_ctzll(same)) + rank;
            } else {
                pos = atomicAdd(&lds[slot], 1u);
            }
            if (SCATTER) idx[pos] = (uint32_t)(v * 16 + 2 * j + hi) | ((code & 0x8000u) ? NEG_BIT : 0u);
        }
        v = vn;
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < range; t += blockDim.x) counts[(gbase + t) * MSM_SLICES + sl] = lds[t];
    }
}

// ---- merged-window sort (SRS bases with window tables) -------------------------------------------
// Window bits C up to 22: digit magnitudes no longer fit the u16 codes of the per-window path.
// code = 0xFFFFFFFF for a zero digit, else bit 31 = sign, bits 0..21 = |d| - 1 (the bucket).
constexpr int MSM_M_MIN_C = 8, MSM_M_MAX_C = 22;
constexpr uint32_t MSM_M_MAX_BINS = 1u << (MSM_M_MAX_C - 1 - MSM_RANGE_MAX_BITS);     // 1024 partitions of 2048 buckets
constexpr uint32_t MSM_M_CHUNK = 2048;                                                   // scalars per workgroup of the partition passes
constexpr uint32_t CODE_ZERO = 0xFFFFFFFFu;
// The top window holds only the few leading bits of a scalar (14 of 20 at C = 20): its digits would all land in the
// lowest buckets -- a handful of partitions with ten times the entries of the others.  Its digit is therefore scaled
// by 2^top_shift (still at most 2^(C-1), so it never goes negative) against a table entry built with top_shift fewer
// doublings: same product, buckets spread over the whole range.
template <int C>
__device__ __forceinline__ void recode_wide(const Fr& s, uint32_t (&code)[(256 + C - 1) / C], int top_shift) {
    constexpr int W = (256 + C - 1) / C;
    constexpr uint32_t mask = (1u << C) - 1, half = 1u << (C - 1);
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int bit = w * C, limb = bit >> 5, sh = bit & 31;
        uint32_t d = limb < 8 ? (s.l[limb < 8 ? limb : 7] >> sh) : 0u;
        if (sh + C > 32 && limb + 1 < 8) d |= s.l[limb + 1 < 8 ? limb + 1 : 7] << (32 - sh);
        d = (d & mask) + carry;
        if (w == W - 1) d <<= top_shift;
        if (d > half) { carry = 1; const uint32_t mag = (1u << C) - d; code[w] = mag ? (NEG_BIT | (mag - 1)) : CODE_ZERO; }
        else { carry = 0; code[w] = d ? (d - 1) : CODE_ZERO; }
    }
}
// LDS counter update shared by the partition and the per-partition passes: lanes of the wave that
// hit the leader's slot share ONE atomic (runs of equal scalars put the same bucket in every lane,
// and 64 atomics on one LDS word serialise); the rest go alone.  Returns the lane's position.
__device__ __forceinline__ uint32_t lds_take(uint32_t* lds, uint32_t slot, bool active) {
    const uint64_t act = __ballot(active);
    if (!active) return 0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lead = __shfl(slot, (int)__builtin_ctzll(act));
    const uint64_t same = __ballot(slot == lead);          // only active lanes reach this ballot
    if (slot == lead) {
        const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        uint32_t pos = 0;
        if (rank == 0) pos = atomicAdd(&lds[slot], (uint32_t)__popcll(same));
        return __shfl(pos, (int)__builtin_ctzll(same)) + rank;
    }
    return atomicAdd(&lds[slot], 1u);
}
// Partition passes.  Workgroup g owns scalars [g * CHUNK, (g + 1) * CHUNK) and recodes them twice:
//   COUNT   hist[bin * nwg + g] = entries of this chunk whose bucket falls into partition `bin`
//   SCATTER the same entries go to entries[cursor++], cursors starting at the scanned histogram:
//           entry = (bucket & (2^range_bits - 1)) << 32 | sign | (w * tab_stride + i)
// (recoding again costs one Montgomery product per scalar; keeping the digits would cost a write
// and a read of W words per scalar).
// LOWPART (the GM sort): a bucket's partition is chosen by the LOW bits of its index and its place inside the partition by the high
// bits.  Small-valued columns put most of their entries into the lowest buckets (bytes: buckets 0..254 of window 0) -- by the
// high bits that is ONE partition per column, a quarter of a million entries streamed by a single workgroup of the counting sort
// while the other 2047 hold a few thousand each (0.6 ms per group of eight columns against 0.17 ms when the values are spread).
// The sort then sees bucket ids with the two bit fields swapped; the accumulation puts them back when it writes a bucket (perm_true).
template <int C, bool SCATTER, bool LOWPART = false>
__device__ __forceinline__ void partition_body(const Fr* __restrict__ scalars, uint64_t n, int range_bits, uint32_t* __restrict__ hist, const uint32_t* __restrict__ hist_off,
                                               uint64_t* __restrict__ entries, uint64_t tab_stride, int top_shift, uint32_t bin_base, uint32_t set_mask, uint32_t nwg, uint32_t g, uint32_t* lds) {
    // grouped columns (GM sort): this column's partitions start at bin_base in the shared histogram, and its windows are dealt over
    // set_mask + 1 bucket sets (window w -> set w & set_mask, key = set * 2^(C-1) + bucket): more, shorter buckets
    constexpr int W = (256 + C - 1) / C;
    const uint32_t nbins = (set_mask + 1u) << (C - 1 - range_bits);
    for (uint32_t t = threadIdx.x; t < nbins; t += blockDim.x) lds[t] = SCATTER ? hist_off[(uint64_t)(bin_base + t) * nwg + g] : 0u;
    __syncthreads();
    const uint64_t chunk = (n + nwg - 1) / nwg;                    // scalars per workgroup (the host sizes the grid)
    const uint64_t lo = min(n, (uint64_t)g * chunk), hi = min(n, lo + chunk);
    const uint32_t rmask = (1u << range_bits) - 1u;
    for (uint64_t base = lo; base < hi; base += blockDim.x) {          // uniform trip count: the ballots below see whole waves
        const uint64_t i = base + threadIdx.x;
        const bool live = i < hi;
        uint32_t code[W];
        if (live) recode_wide<C>(from_mont(ldg(scalars + i)), code, top_shift);
        else {
#pragma unroll
            for (int w = 0; w < W; ++w) code[w] = CODE_ZERO;
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const bool nz = code[w] != CODE_ZERO;
            if (!__ballot(nz)) continue;
            uint32_t part, local;
            if (LOWPART) {
                const int PB = C - 1 - range_bits;                     // partition bits per bucket set
                const uint32_t b = code[w] & 0x3FFFFFu;
                part = ((((uint32_t)w & set_mask)) << PB) | (b & ((1u << PB) - 1u));
                local = b >> PB;
            } else {
                const uint32_t bucket = (code[w] & 0x3FFFFFu) | (((uint32_t)w & set_mask) << (C - 1));
                part = bucket >> range_bits;
                local = bucket & rmask;
            }
            const uint32_t pos = lds_take(lds, part, nz);
            if (SCATTER && nz) entries[pos] = ((uint64_t)local << 32) | (uint64_t)((uint32_t)((uint64_t)w * tab_stride + i) | (code[w] & NEG_BIT));
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < nbins; t += blockDim.x) hist[(uint64_t)(bin_base + t) * nwg + g] = lds[t];
    }
}
template <int C, bool SCATTER>
__global__ void __launch_bounds__(256) k_msm_m_partition(const Fr* __restrict__ scalars_arg, uint64_t n, int range_bits, uint32_t* __restrict__ hist, const uint32_t* __restrict__ hist_off,
                                                          uint64_t* __restrict__ entries, uint64_t tab_stride, int top_shift, const MsmCol* __restrict__ cols = nullptr, const uint32_t* __restrict__ ctr = nullptr) {
    const Fr* __restrict__ scalars = cols ? cols[*ctr].scalars : scalars_arg;
    __shared__ uint32_t lds[MSM_M_MAX_BINS];
    partition_body<C, SCATTER>(scalars, n, range_bits, hist, hist_off, entries, tab_stride, top_shift, 0u, 0u, gridDim.x, blockIdx.x, lds);
}
// The same pass over the columns of a GM group in ONE launch (blockIdx.y = column of the group): a column's 512 workgroups of 256
// threads leave three quarters of the device's wave slots empty; eight columns fill them.
constexpr int GM_MAX_COLS = 16;
struct GmCols { const Fr* p[GM_MAX_COLS]; };
template <int C, bool SCATTER>
__global__ void __launch_bounds__(256) k_msm_gm_partition(GmCols cols, uint64_t n, int range_bits, uint32_t* __restrict__ hist, const uint32_t* __restrict__ hist_off,
                                                           uint64_t* __restrict__ entries, uint64_t tab_stride, uint32_t bins_per_col, uint32_t set_mask) {
    __shared__ uint32_t lds[MSM_M_MAX_BINS];
    partition_body<C, SCATTER, true>(cols.p[blockIdx.y], n, range_bits, hist, hist_off, entries, tab_stride, 0, blockIdx.y * bins_per_col, set_mask, gridDim.x, blockIdx.x, lds);
}
// Scatter pass of the partition step with the runs staged in LDS.  k_msm_m_partition<C, true> lets every lane
// write its 8-byte entry to the cursor of its own partition: 64 lanes, 64 partitions, 64 separate 32-byte
// sectors -- 343 MiB written for 104 MiB of entries.  Here a workgroup (one scalar per thread) ranks its entries
// per partition in LDS, lays them out partition by partition in a staging buffer and copies the buffer out
// with consecutive lanes on consecutive entries: every (workgroup, partition) run leaves as one contiguous
// burst.  Dynamic LDS: cnt[1024] | gdelta[1024] | wtot[16] | stage[1024 W] (u64) | pid[1024 W] (u16).
constexpr uint32_t MSM_M_SCHUNK = 1024;                 // scalars per workgroup of the staged scatter (and of its histogram pass)
template <int C>
__global__ void __launch_bounds__(1024) k_msm_m_scatter_staged(const Fr* __restrict__ scalars_arg, uint64_t n, int range_bits, const uint32_t* __restrict__ hist_off,
                                                               uint64_t* __restrict__ entries, uint64_t tab_stride, int top_shift, const MsmCol* __restrict__ cols = nullptr, const uint32_t* __restrict__ ctr = nullptr) {
    constexpr int W = (256 + C - 1) / C;
    const Fr* __restrict__ scalars = cols ? cols[*ctr].scalars : scalars_arg;
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    uint32_t* cnt = sm;                                  // counters, then local start of every partition
    uint32_t* gdelta = sm + MSM_M_MAX_BINS;              // global cursor of (partition, this workgroup) minus the local start
    uint32_t* wtot = gdelta + MSM_M_MAX_BINS;
    uint64_t* stage = reinterpret_cast<uint64_t*>(wtot + 16);
    uint16_t* pid = reinterpret_cast<uint16_t*>(stage + (size_t)MSM_M_SCHUNK * W);
    const uint32_t nbins = 1u << (C - 1 - range_bits), nwg = gridDim.x, g = blockIdx.x;
    const uint32_t rmask = (1u << range_bits) - 1u;
    cnt[threadIdx.x] = 0u;                               // blockDim.x == MSM_M_MAX_BINS == 1024
    __syncthreads();
    const uint64_t chunk = (n + nwg - 1) / nwg;          // <= MSM_M_SCHUNK (the host sizes the grid)
    const uint64_t i = min(n, (uint64_t)g * chunk) + threadIdx.x;
    const bool live = threadIdx.x < chunk && i < min(n, ((uint64_t)g + 1) * chunk);
    uint32_t code[W], rank[W];
    if (live) recode_wide<C>(from_mont(ldg(scalars + i)), code, top_shift);
    else {
#pragma unroll
        for (int w = 0; w < W; ++w) code[w] = CODE_ZERO;
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const bool nz = code[w] != CODE_ZERO;
        rank[w] = 0;
        if (!__ballot(nz)) continue;
        rank[w] = lds_take(cnt, (code[w] & 0x3FFFFFu) >> range_bits, nz);
    }
    __syncthreads();
    {   // exclusive scan of the partition counters, one per thread
        const uint32_t c0 = threadIdx.x < nbins ? cnt[threadIdx.x] : 0u;
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        uint32_t incl = c0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if (lane >= (uint32_t)off) incl += o; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < wave; ++w) wbase += wtot[w];
        const uint32_t ex = wbase + incl - c0;
        if (threadIdx.x < nbins) {
            cnt[threadIdx.x] = ex;
            gdelta[threadIdx.x] = hist_off[(uint64_t)threadIdx.x * nwg + g] - ex;
        }
    }
    __syncthreads();
    uint32_t total = 0;
    for (uint32_t w = 0; w < 16; ++w) total += wtot[w];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        if (code[w] == CODE_ZERO) continue;
        const uint32_t bucket = code[w] & 0x3FFFFFu, part = bucket >> range_bits;
        const uint32_t pos = cnt[part] + rank[w];
        stage[pos] = ((uint64_t)(bucket & rmask) << 32) | (uint64_t)((uint32_t)((uint64_t)w * tab_stride + i) | (code[w] & NEG_BIT));
        pid[pos] = (uint16_t)part;
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) entries[gdelta[pid[e]] + e] = stage[e];
}
static size_t scatter_staged_lds(int W) { return (size_t)(2 * MSM_M_MAX_BINS + 16) * 4 + (size_t)MSM_M_SCHUNK * W * 10; }
// Per-partition LDS counting sort: workgroup (bin, slice) streams its quarter of the partition's
// entries; counters are laid out [bucket][slice] exactly as in the per-window path, so the same
// scans produce every (bucket, slice) cursor.
template <bool SCATTER>
__global__ void __launch_bounds__(1024) k_msm_m_bin(const uint64_t* __restrict__ entries, const uint32_t* __restrict__ hist_off, uint32_t nwg, int range_bits,
                                                    uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets, uint32_t* __restrict__ idx) {
    __shared__ uint32_t lds[1 << MSM_RANGE_MAX_BITS];
    // Workgroups go to the 8 XCDs round-robin by linear id.  The MSM_SLICES workgroups of a partition write
    // interleaved 4-byte runs into the same lines of `idx` (cursors are laid out [bucket][slice]); placed on
    // ONE XCD they merge those lines in its L2 instead of evicting four partial copies (the L2s are not
    // coherent with each other): id = xcd + 8 * (MSM_SLICES * group + slice), partition = 8 * group + xcd.
    const uint32_t nbins_ = gridDim.x / MSM_SLICES;
    uint32_t bin, sl;
    if ((nbins_ & 7u) == 0) {
        const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        sl = slot % MSM_SLICES;
        bin = (slot / MSM_SLICES) * 8u + xcd;
    } else {
        bin = blockIdx.x / MSM_SLICES;
        sl = blockIdx.x % MSM_SLICES;
    }
    const uint32_t range = 1u << range_bits;
    const uint64_t gbase = (uint64_t)bin << range_bits;
    for (uint32_t t = threadIdx.x; t < range; t += blockDim.x) lds[t] = SCATTER ? offsets[(gbase + t) * MSM_SLICES + sl] : 0u;
    __syncthreads();
    const uint32_t lo = hist_off[(uint64_t)bin * nwg], hi = hist_off[(uint64_t)(bin + 1) * nwg];     // the scan's closing entry holds the total
    const uint32_t per = (hi - lo + MSM_SLICES - 1) / MSM_SLICES;
    const uint32_t s0 = min(hi, lo + sl * per), s1 = min(hi, s0 + per);
    for (uint32_t base = s0; base < s1; base += blockDim.x) {
        const uint32_t e = base + threadIdx.x;
        const bool live = e < s1;
        const uint64_t ent = live ? entries[e] : 0ull;
        const uint32_t pos = lds_take(lds, (uint32_t)(ent >> 32), live);
        if (SCATTER && live) idx[pos] = (uint32_t)ent;
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < range; t += blockDim.x) counts[(gbase + t) * MSM_SLICES + sl] = lds[t];
    }
}

// One-launch counting sort of a partition (the usual case: the partition's indices fit the LDS staging
// buffer).  Workgroup `bin` counts its entries per bucket, scans the 2^range_bits counters in LDS --
// partitions are contiguous in the index array, so offsets[b] = partition start + local prefix needs no
// global scan --, then ranks the entries again (the second read comes from L2 / MALL), places the 4-byte
// table indices in LDS and writes them out as ONE contiguous run: no partial-line writes (the four-slice
// scatter above wrote 10x its payload through L2).  Also leaves counts[], the size histogram and the
// closing offsets[nb] that the task split reads.  A partition larger than the staging buffer (skewed
// scalars) scatters straight to global memory instead.
constexpr uint32_t MSM_M_STAGE = 15u * 1024u;      // 60 KiB of indices + 8 KiB of counters: two workgroups per CU
constexpr uint32_t MSM_SIZE_BINS = 256;            // == SIZE_BINS below
__global__ void __launch_bounds__(1024) k_msm_m_binsort(const uint64_t* __restrict__ entries, const uint32_t* __restrict__ hist_off, uint32_t nwg, int range_bits, uint32_t nb,
                                                        uint32_t* __restrict__ offsets, uint32_t* __restrict__ counts, uint32_t* __restrict__ size_hist, uint32_t* __restrict__ idx) {
    __shared__ uint32_t cnt[1 << MSM_RANGE_MAX_BITS];
    __shared__ uint32_t stage[MSM_M_STAGE];
    __shared__ uint32_t lh[MSM_SIZE_BINS];
    __shared__ uint32_t wtot[16];
    const uint32_t bin = blockIdx.x, range = 1u << range_bits;
    const uint64_t gbase = (uint64_t)bin << range_bits;
    for (uint32_t t = threadIdx.x; t < range; t += blockDim.x) cnt[t] = 0u;
    if (threadIdx.x < MSM_SIZE_BINS) lh[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t lo = hist_off[(uint64_t)bin * nwg], hi = hist_off[(uint64_t)(bin + 1) * nwg];     // the scan's closing entry holds the total
    const bool staged = hi - lo <= MSM_M_STAGE;
    // a staged partition is read ONCE: every thread keeps its <= 15 entries in registers between the counting and the
    // ranking pass, all loads in flight together (one latency, not fifteen)
    constexpr int PER = MSM_M_STAGE / 1024;
    uint64_t ent_r[PER];
    if (staged) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t e = lo + (uint32_t)j * 1024u + threadIdx.x;
            ent_r[j] = e < hi ? entries[e] : ~0ull;                      // no real entry is all ones (its high word is a bucket index)
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (lo + (uint32_t)j * 1024u >= hi) break;                       // uniform
            const bool live = ent_r[j] != ~0ull;
            (void)lds_take(cnt, (uint32_t)(ent_r[j] >> 32), live);
        }
    } else {
        // an oversize partition (skewed scalars: e.g. the carry digit of every small value >= 2^(c-1) lands in ONE bucket of the next
        // window -- 150 000 entries in one partition of a witness-like column) is streamed by this one workgroup: eight loads in
        // flight per trip, or its two passes are a chain of a few hundred load latencies that the whole launch waits for
        constexpr int UNR = 8;
        for (uint32_t base = lo; base < hi; base += blockDim.x * UNR) {
            uint64_t ent[UNR];
#pragma unroll
            for (int j = 0; j < UNR; ++j) { const uint32_t e = base + (uint32_t)j * blockDim.x + threadIdx.x; ent[j] = e < hi ? entries[e] : ~0ull; }
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                if (base + (uint32_t)j * blockDim.x >= hi) break;                     // uniform
                (void)lds_take(cnt, (uint32_t)(ent[j] >> 32), ent[j] != ~0ull);
            }
        }
    }
    __syncthreads();
    {   // exclusive scan of the counters: thread t owns buckets 2t, 2t + 1 (range <= 2048)
        const uint32_t t2 = 2 * threadIdx.x;
        const uint32_t c0 = t2 < range ? cnt[t2] : 0u, c1 = t2 + 1 < range ? cnt[t2 + 1] : 0u;
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        uint32_t incl = c0 + c1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if (lane >= (uint32_t)off) incl += o; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < wave; ++w) wbase += wtot[w];
        const uint32_t ex = wbase + incl - (c0 + c1);
        if (t2 < range) {
            cnt[t2] = ex;
            offsets[gbase + t2] = lo + ex;
            counts[gbase + t2] = c0;
            atomicAdd(&lh[min(c0, MSM_SIZE_BINS - 1)], 1u);
        }
        if (t2 + 1 < range) {
            cnt[t2 + 1] = ex + c0;
            offsets[gbase + t2 + 1] = lo + ex + c0;
            counts[gbase + t2 + 1] = c1;
            atomicAdd(&lh[min(c1, MSM_SIZE_BINS - 1)], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < MSM_SIZE_BINS && lh[threadIdx.x]) atomicAdd(&size_hist[threadIdx.x], lh[threadIdx.x]);
    if (bin + 1 == gridDim.x && threadIdx.x == 0) offsets[nb] = hi;
    if (staged) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (lo + (uint32_t)j * 1024u >= hi) break;                       // uniform
            const bool live = ent_r[j] != ~0ull;
            const uint32_t pos = lds_take(cnt, (uint32_t)(ent_r[j] >> 32), live);
            if (live) stage[pos] = (uint32_t)ent_r[j];
        }
    } else {
        constexpr int UNR = 8;
        for (uint32_t base = lo; base < hi; base += blockDim.x * UNR) {
            uint64_t ent[UNR];
#pragma unroll
            for (int j = 0; j < UNR; ++j) { const uint32_t e = base + (uint32_t)j * blockDim.x + threadIdx.x; ent[j] = e < hi ? entries[e] : ~0ull; }
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                if (base + (uint32_t)j * blockDim.x >= hi) break;                     // uniform
                const bool live = ent[j] != ~0ull;
                const uint32_t pos = lds_take(cnt, (uint32_t)(ent[j] >> 32), live);
                if (live) idx[lo + pos] = (uint32_t)ent[j];
            }
        }
    }
    if (staged) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < hi - lo; t += blockDim.x) idx[lo + t] = stage[t];
    }
}

// ---- exclusive scan of the bucket counts, three small kernels (4096 counts per block) ----------
constexpr int SCAN_T = 1024, SCAN_ITEMS = 4;
__device__ __forceinline__ uint32_t block_scan_u32(uint32_t v, uint32_t* sh, uint32_t* total) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_T; off <<= 1) {
        uint32_t x = sh[threadIdx.x];
        if ((int)threadIdx.x >= off) x += sh[threadIdx.x - off];
        __syncthreads();
        sh[threadIdx.x] = x;
        __syncthreads();
    }
    *total = sh[SCAN_T - 1];
    const uint32_t ex = threadIdx.x ? sh[threadIdx.x - 1] : 0u;
    __syncthreads();
    return ex;
}
__global__ void __launch_bounds__(SCAN_T) k_scan_u32_a(const uint32_t* __restrict__ counts, uint32_t cnt, uint32_t* __restrict__ offsets, uint32_t* __restrict__ block_tot) {
    __shared__ uint32_t sh[SCAN_T];
    const uint32_t base = blockIdx.x * (SCAN_T * SCAN_ITEMS) + threadIdx.x * SCAN_ITEMS;
    uint32_t c[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { c[k] = base + k < cnt ? counts[base + k] : 0u; sum += c[k]; }
    uint32_t total;
    uint32_t run = block_scan_u32(sum, sh, &total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < cnt) offsets[base + k] = run; run += c[k]; }
    if (threadIdx.x == 0) block_tot[blockIdx.x] = total;
}
__global__ void __launch_bounds__(SCAN_T) k_scan_u32_b(uint32_t* block_tot, uint32_t nblocks, uint32_t* offsets, uint32_t cnt, uint32_t* total_out) {
    __shared__ uint32_t sh[SCAN_T];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += SCAN_T) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? block_tot[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_scan_u32(v, sh, &total);
        if (i < nblocks) block_tot[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) { offsets[cnt] = carry; if (total_out) *total_out = carry; }
}
// Finishes the scan over the [bucket][slice] counters (adds the block offsets: slice_off = cursor
// start of every (bucket, slice)), derives the per-bucket view (offsets[b] = slice_off[b][0],
// counts[b] = sum of its slices) and bins every bucket by size (descending order of size -> a wave
// works on buckets of equal length; big ones start first).
constexpr uint32_t SIZE_BINS = 256;
constexpr uint32_t MSM_WFLAGS = 256;           // window flags of one sort (behind nmulti + 4): up to eight small-valued columns of 16 windows share a launch sequence
constexpr uint32_t TASK_DONE_MAX = 1u << 17;   // done-counters (behind nmulti + 4 + MSM_WFLAGS) of the split buckets that straddle a wave boundary inside k_msm_buckets: positions below this
constexpr uint32_t TASK_INLINE_MAX = 8;    // ... when they were split into at most this many tasks
constexpr uint32_t TASK_CAP = 48;       // points per task, see "skew-proof work split" below
__global__ void __launch_bounds__(SCAN_T) k_scan_u32_c(const uint32_t* __restrict__ slice_counts, uint32_t nbuckets, uint32_t* __restrict__ slice_off, const uint32_t* __restrict__ block_tot,
                                                       uint32_t* __restrict__ offsets, uint32_t* __restrict__ counts, uint32_t* __restrict__ size_hist) {
    __shared__ uint32_t lh[SIZE_BINS];
    if (threadIdx.x < SIZE_BINS) lh[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t off = block_tot[blockIdx.x];
    const uint32_t b = blockIdx.x * SCAN_T + threadIdx.x;        // SCAN_ITEMS == MSM_SLICES entries per bucket, one bucket per thread
    if (b < nbuckets) {
        const uint4 c = reinterpret_cast<const uint4*>(slice_counts)[b];
        uint4 o = reinterpret_cast<const uint4*>(slice_off)[b];
        o.x += off; o.y += off; o.z += off; o.w += off;
        reinterpret_cast<uint4*>(slice_off)[b] = o;
        const uint32_t total = c.x + c.y + c.z + c.w;
        offsets[b] = o.x;
        counts[b] = total;
        atomicAdd(&lh[min(total, SIZE_BINS - 1)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < SIZE_BINS && lh[threadIdx.x]) atomicAdd(&size_hist[threadIdx.x], lh[threadIdx.x]);
}
static_assert(MSM_SIZE_BINS == SIZE_BINS, "k_msm_m_binsort bins bucket sizes like k_scan_u32_c");
static_assert(SCAN_ITEMS == MSM_SLICES, "k_scan_u32_a scans MSM_SLICES entries per thread: one bucket");
// size_hist (256 bins) -> start offset of each bin in the descending-size order
// One wave.  Also picks this MSM's task size: TASK_CAP points when there is plenty of work, smaller
// when the scalars fill only a few windows (witness columns: values below 2^64 leave 12 of 16
// windows empty, and one lane per 32-point bucket would be 2 waves per SIMD) -- so that about
// 2^18 tasks exist either way.  nmulti[0] = M = buckets with more points than one task,
// nmulti[1] = the task size.
constexpr uint32_t TASK_MIN = 8, TASK_TARGET = 1u << 18;
#ifndef ZK_TASK_TMIN
#define ZK_TASK_TMIN (1u << 16)
#endif
__global__ void __launch_bounds__(64) k_size_bins_scan(uint32_t* size_hist, const uint32_t* __restrict__ total_points) {
    const uint32_t lane = threadIdx.x;
    const uint32_t total = *total_points;
    uint32_t cap = TASK_CAP;
    if (total < TASK_CAP * TASK_TARGET) cap = max(TASK_MIN, (total + TASK_TARGET - 1) / TASK_TARGET);
    // ... unless whole buckets already make enough tasks.  A dense column at 2^18 (3.9 M entries over 2^17 buckets of ~30) got task
    // size 15 from the rule above: EVERY bucket split in two or three, and the combination kernels behind the accumulation took
    // longer than the accumulation itself (0.47 against 0.40 ms per MSM, kernel statistics of tools/msm_graph_pipes.py).  When
    // tasks of TASK_CAP points -- one per bucket for such a column -- still number ZK_TASK_TMIN, they win: 2^18 dense 0.61 -> 0.55 ms,
    // 2^19 0.91 -> 0.77 ms per MSM in a batch.  A column of bits (a few giant buckets) or a 2^16 column keeps the small tasks
    // (an in-between size -- every bucket split AND fewer lanes -- measured worse there: 0.24 -> 0.27 ms).
    if ((ZK_TASK_TMIN) != 0u && cap < TASK_CAP) {
        uint32_t tasks = 0;
        for (uint32_t bin = 1 + lane; bin < SIZE_BINS; bin += 64) tasks += size_hist[bin] * ((bin + TASK_CAP - 1) / TASK_CAP);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tasks += __shfl_down(tasks, off);
        if (__shfl(tasks, 0) >= (uint32_t)(ZK_TASK_TMIN)) cap = TASK_CAP;
    }
    // descending order: position p = SIZE_BINS - 1 - bin; lane owns positions 4 lane .. 4 lane + 3
    uint32_t c[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = size_hist[SIZE_BINS - 1 - (4 * lane + k)]; sum += c[k]; }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if (lane >= (uint32_t)off) incl += o; }
    uint32_t run = incl - sum;
    // lanes (= tasks) of split buckets against all lanes: k_msm_buckets puts split buckets together itself only when they are a
    // small part of the launch (a dense column: 7 %).  When every bucket is split (a column of small values: few entries, task
    // size 8, all 262 144 lanes are tasks) the segmented reduction would run on every wave; the combination kernels do that
    // work better (measured: 0.41 against 0.17 + 0.14 ms for 30-bit values, tools/msm_narrow.py) -- then H = M: all heavy.
    uint32_t t_split = 0, t_all = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t bin = SIZE_BINS - 1 - (4 * lane + k);
        const uint32_t per = bin ? (bin + cap - 1) / cap : 1u;
        t_all += c[k] * per;
        if (bin > cap) t_split += c[k] * per;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { t_split += __shfl_down(t_split, off); t_all += __shfl_down(t_all, off); }
    t_split = __shfl(t_split, 0);
    t_all = __shfl(t_all, 0);
    const bool inline_ok = (uint64_t)t_split * 8u <= (uint64_t)t_all;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t bin = SIZE_BINS - 1 - (4 * lane + k);
        if (bin == cap) { size_hist[SIZE_BINS] = run; if (!inline_ok) size_hist[SIZE_BINS + 2] = run; }          // M: buckets with more than `cap` points come first
        // H: the "heavy" ones among them, split into more than TASK_INLINE_MAX tasks (or of unknown size: the last bin); they
        // come first of all.  Buckets at positions [H, M) are put together inside k_msm_buckets.
        if (inline_ok && bin == min(TASK_INLINE_MAX * cap, SIZE_BINS - 2)) size_hist[SIZE_BINS + 2] = run;
        size_hist[bin] = run;
        run += c[k];
    }
    if (lane == 0) size_hist[SIZE_BINS + 1] = cap;
}
// ---- skew-proof work split ------------------------------------------------------------------------
// A bucket with c points becomes ceil(c / cap) tasks of <= cap consecutive points (cap <= TASK_CAP,
// chosen per MSM by k_size_bins_scan), so no lane ever walks more than TASK_CAP points whatever the scalar distribution (selector / boolean /
// small-value columns put n/2 points into one bucket).  Tasks are numbered along the size-ordered
// bucket sequence (a wave still sees equal-length work); single-task buckets write their bucket
// directly, multi-task buckets write partials that one workgroup per bucket tree-sums afterwards.

// order[pos] = bucket id, grouped by size bin (block-aggregated reservation of output ranges);
// ntasks[pos] = number of tasks of that bucket
__global__ void __launch_bounds__(SCAN_T) k_order_buckets(const uint32_t* __restrict__ counts, uint32_t cnt, uint32_t* __restrict__ size_cursor, uint32_t* __restrict__ order,
                                                          uint32_t* __restrict__ ntasks) {
    __shared__ uint32_t lh[SIZE_BINS], lbase[SIZE_BINS];
    if (threadIdx.x < SIZE_BINS) lh[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t cap = size_cursor[SIZE_BINS + 1];          // this MSM's task size (k_size_bins_scan)
    const uint32_t base = blockIdx.x * (SCAN_T * SCAN_ITEMS);
    uint32_t bin[SCAN_ITEMS], rank[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t i = base + k * SCAN_T + threadIdx.x;
        bin[k] = i < cnt ? min(counts[i], SIZE_BINS - 1) : 0xffffffffu;
        if (i < cnt) rank[k] = atomicAdd(&lh[bin[k]], 1u);
    }
    __syncthreads();
    if (threadIdx.x < SIZE_BINS && lh[threadIdx.x]) lbase[threadIdx.x] = atomicAdd(&size_cursor[threadIdx.x], lh[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t i = base + k * SCAN_T + threadIdx.x;
        if (i < cnt) {
            const uint32_t pos = lbase[bin[k]] + rank[k];
            order[pos] = i;
            ntasks[pos] = (counts[i] + cap - 1) / cap;
        }
    }
}
// adds the block offsets of the task scan
__global__ void __launch_bounds__(SCAN_T) k_task_offsets(uint32_t cnt, uint32_t* __restrict__ toff, const uint32_t* __restrict__ block_tot) {
    const uint32_t off = block_tot[blockIdx.x];
    const uint32_t base = blockIdx.x * (SCAN_T * SCAN_ITEMS);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t i = base + k * SCAN_T + threadIdx.x;
        if (i < cnt) toff[i] += off;
    }
}

// bases (canonical, R = 2^256 form) -> R' = 2^261 form (x32 mod p), identity preserved
__global__ void k_bases_to_rprime(const G1Affine* __restrict__ in, G1Affine* __restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine p = ldg(in + i);
#pragma unroll
    for (int k = 0; k < 5; ++k) { p.x = dbl(p.x); p.y = dbl(p.y); }
    stg(out + i, p);
}

// Fixed-base window tables (SRS bases only): table[w][i] = 2^(c*w) * P_i, affine, R' form.  With
// them every window's bucket b carries the same weight (b+1): the W per-window bucket arrays are
// first folded into one, the weighted reduction runs over 2^(c-1) buckets instead of W * 2^(c-1),
// and the host Horner tail disappears.
__global__ void __launch_bounds__(256) k_build_window_tables(const G1Affine* __restrict__ bases_rp, uint64_t n, int c, int W, G1Affine* __restrict__ table, int top_shift) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine p0 = ldg(bases_rp + i);
    stg(table + i, p0);
    G1Xyzz29 cur = p0.is_identity() ? identity29() : G1Xyzz29{unpack29<Fq29P>(p0.x), unpack29<Fq29P>(p0.y), one29(), one29()};
#pragma unroll 1
    for (int w = 1; w < W; ++w) {
        const int steps = w == W - 1 ? c - top_shift : c;        // merged-window plans scale the top digit instead (MsmPlan::top_shift)
#pragma unroll 1
        for (int j = 0; j < steps; ++j) cur = dbl29pt(cur);
        const G1Affine a = to_affine_rp(cur);
        stg(table + (uint64_t)w * n + i, a);
        if (!a.is_identity()) cur = G1Xyzz29{unpack29<Fq29P>(a.x), unpack29<Fq29P>(a.y), one29(), one29()};   // keep Z = 1: cheaper doublings stay exact
    }
}
// folded[b] = sum_w buckets[w*B + b]: four lanes per bucket (each takes every 4th window), LDS combine
__global__ void __launch_bounds__(256) k_msm_fold_windows(const G1Xyzz29* __restrict__ buckets, uint32_t B, int W, G1Xyzz29* __restrict__ folded, const uint32_t* __restrict__ wflag) {
    __shared__ G1Xyzz29 sh[256];
    buckets += (uint64_t)blockIdx.y * W * B;          // several columns per launch: column y owns windows [y W, (y + 1) W) and folded[y B, (y + 1) B)
    wflag += blockIdx.y * W;
    folded += (uint64_t)blockIdx.y * B;
    const uint32_t b = blockIdx.x * 64 + (threadIdx.x >> 2), q = threadIdx.x & 3;
    G1Xyzz29 acc = identity29();
    if (b < B) for (int w = (int)q; w < W; w += 4) if (wflag[w]) acc = add29pt(acc, ldg29(buckets + (uint64_t)w * B + b));   // empty windows were never written
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (q < 2) sh[threadIdx.x] = add29pt(sh[threadIdx.x], sh[threadIdx.x + 2]);
    __syncthreads();
    if (q == 0 && b < B) stg29(folded + b, add29pt(sh[threadIdx.x], sh[threadIdx.x + 1]));
}

__device__ __forceinline__ G1Xyzz29 shfl_down_pt(const G1Xyzz29& p, int off) {
    G1Xyzz29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.x.l[i] = __shfl_down(p.x.l[i], off);
        r.y.l[i] = __shfl_down(p.y.l[i], off);
        r.zz.l[i] = __shfl_down(p.zz.l[i], off);
        r.zzz.l[i] = __shfl_down(p.zzz.l[i], off);
    }
    return r;
}
__device__ __forceinline__ G1Xyzz29 accumulate_run(const G1Affine* __restrict__ bases_rp, const uint32_t* __restrict__ idx, uint32_t lo, uint32_t hi) {
    G1Xyzz29 acc = identity29();
    for (uint32_t j = lo; j < hi; ++j) {
        const uint32_t v = idx[j];
        G1Affine29 q = load_affine29(bases_rp + (v & ~NEG_BIT));
        if ((v & NEG_BIT) && !is_identity29(q)) q.y = neg_canon29(q.y);
        acc = madd29(acc, q);
    }
    return acc;
}
// Bucket id as the GM sort numbers it (low bits of the index first: see partition_body's LOWPART) -> the bucket's true position.
// perm = (RB << 8) | PB, a set holds 2^(PB + RB) buckets; 0 = the sort's numbering is the true one.
__device__ __forceinline__ uint32_t perm_true(uint32_t b, uint32_t perm) {
    if (!perm) return b;
    const uint32_t PB = perm & 0xffu, RB = perm >> 8, in_set = (1u << (PB + RB)) - 1u;
    const uint32_t low = (b & in_set) >> RB, t = b & ((1u << RB) - 1u);
    return (b & ~in_set) | (t << PB) | low;
}
// Buckets are ordered by decreasing size, so the M = *nmulti buckets with more than TASK_CAP
// points are exactly positions [0, M).  Single-task (and empty) buckets: one lane per position,
// no search, result straight into the bucket.
// One launch, virtual index v: the Tm = toff[M] tasks of the multi-task buckets come first (they
// are the long poles and overlap with everything behind them), then one lane per ordinary bucket
// position M + (v - Tm).  toff = exclusive scan of ntasks over positions.
__global__ void __launch_bounds__(256) k_msm_buckets(const G1Affine* __restrict__ bases_rp, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ idx,
                                                     const uint32_t* __restrict__ order, const uint32_t* __restrict__ toff, const uint32_t* __restrict__ nmulti,
                                                     uint32_t nbuckets, G1Xyzz29* __restrict__ buckets, G1Xyzz29* __restrict__ partial,
                                                     int log_b, uint64_t tab_stride, const uint32_t* __restrict__ wflag, uint32_t tab_windows = 0, uint32_t perm = 0) {   // tab_stride != 0: bases_rp is a window table, window = bucket >> log_b
                                                     // tab_windows != 0: several columns share the launch, column j owns windows [j W, (j + 1) W): table window = window mod W
    const uint32_t M = *nmulti;
    const uint32_t Tm = M ? toff[M] : 0u;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    // what this lane accumulates: points idx[lo, hi) into bucket b -- ONE call site of the accumulation loop for every kind of
    // lane (2100 instructions of mixed addition: a second inlined copy competes for the instruction cache two CUs share)
    enum : uint32_t { NONE, ORDINARY, HEAVY, IN_WAVE, STRADDLE, LEFT };
    // the task behind virtual index v: position of its bucket, first task of the bucket, task count -- recomputed after the loop
    // instead of kept alive across it (four registers that would push the kernel past 128 VGPRs = from four waves per SIMD to three)
    auto task_of = [&](uint32_t& lo_p, uint32_t& t0, uint32_t& nt) {
        uint32_t hi_p = M;                       // toff[lo_p] <= v < toff[hi_p]
        lo_p = 0;
        while (hi_p - lo_p > 1) {
            const uint32_t mid = (lo_p + hi_p) >> 1;
            if (toff[mid] <= v) lo_p = mid; else hi_p = mid;
        }
        t0 = toff[lo_p];
        nt = toff[lo_p + 1] - t0;
    };
    uint32_t b = 0, lo = 0, hi = 0;
    bool ordinary = false;
    if (v < Tm) {
        uint32_t lo_p, t0, nt;
        task_of(lo_p, t0, nt);
        const uint32_t cap = nmulti[1];
        b = order[lo_p];
        lo = offsets[b] + (v - t0) * cap;
        hi = min(lo + cap, offsets[b + 1]);
    } else {
        const uint32_t p = M + (v - Tm);
        if (p < nbuckets) {
            b = order[p];
            if (!(tab_stride && wflag[b >> log_b] == 0u)) { ordinary = true; lo = offsets[b]; hi = offsets[b + 1]; }      // empty window: the fold skips it, nothing to write
        }
    }
    uint32_t twin = b >> log_b;
    if (tab_windows) twin %= tab_windows;
    G1Xyzz29 acc = accumulate_run(bases_rp + (uint64_t)twin * tab_stride, idx, lo, hi);
    if (ordinary) stg29(buckets + perm_true(b, perm), acc);
    if ((v & ~63u) >= Tm) return;                // a wave of ordinary buckets only: done
    if (nmulti[2] >= M) {                        // every split bucket is left to the combination kernels (k_size_bins_scan decided: too many of them)
        if (v < Tm) stg29(partial + v, acc);
        return;
    }
    // ---- a wave that holds tasks of split buckets (they come first).  A bucket that was split into a handful of tasks -- the
    // 12 000 buckets of the scaled top window of a uniform column (three tasks each), the tail of the size distribution -- is put
    // together HERE: its tasks are consecutive lanes, so a segmented shuffle reduction at the end of the wave (two or three
    // additions) finishes it without touching memory.  Only a bucket whose tasks straddle a wave boundary (~3 %) goes through
    // memory: partials published with a device-scope fence, a counter, and the last arriver adds them up (the fence writes the
    // XCD's dirty L2 lines back: 37 000 of them per launch cost 40 us, profiles/r03_combine.md; a thousand do not).  The
    // combination kernels behind this one keep the "heavy" buckets -- more than TASK_INLINE_MAX tasks: selector / boolean
    // columns -- and exit at once when there are none.
    uint32_t mode = NONE, lo_p = 0, t0 = 0, nt = 0, chunk = 0;
    if (v < Tm) {
        task_of(lo_p, t0, nt);
        chunk = v - t0;
        mode = lo_p < nmulti[2] ? HEAVY : ((lane >= chunk && lane - chunk + nt <= 64u) ? IN_WAVE : (lo_p < TASK_DONE_MAX ? STRADDLE : LEFT));
    }
    if (mode == HEAVY || mode == LEFT) stg29(partial + v, acc);
    if (mode == LEFT && chunk == 0) atomicAdd(const_cast<uint32_t*>(nmulti) + 3, 1u);       // a straddling bucket beyond the done-counters: the combination kernels take it (and must run)
    if (mode == STRADDLE) {
        uint32_t* done = const_cast<uint32_t*>(nmulti) + 4 + MSM_WFLAGS;            // zeroed with the size histogram before every MSM
        stg29(partial + v, acc);
        __threadfence();
        if (atomicAdd(done + lo_p, 1u) + 1u == nt) {
            __threadfence();                     // acquire at device scope: the other tasks may have run behind another XCD's L2
            for (uint32_t i = 0; i < nt; ++i)
                if (i != chunk) acc = add29pt(acc, ldg29(partial + t0 + i));
            stg29(buckets + perm_true(b, perm), acc);
        }
    }
    const uint32_t key = mode == IN_WAVE ? lo_p : 0xFFFFFF00u + lane;      // other lanes: unique keys, never merged
    for (int off = 1; off < (int)TASK_INLINE_MAX; off <<= 1) {
        const uint32_t okey = __shfl_down(key, off);
        const bool take = lane + off < 64u && okey == key;
        if (!__ballot(take)) break;              // segments are contiguous: no pair at this distance, none further
        const G1Xyzz29 other = shfl_down_pt(acc, off);
        if (take) acc = add29pt(acc, other);
    }
    if (mode == IN_WAVE && chunk == 0) stg29(buckets + perm_true(b, perm), acc);
}
// ---- combining the task partials of multi-task buckets ---------------------------------------------
// 1. k_msm_combine_wave: one lane per task partial; lanes of a wave that belong to the same bucket
//    are summed with a segmented shuffle reduction (<= 6 dependent additions), the first lane of
//    every segment writes the sum back.  What is left per bucket are its "leader" slots: its first
//    task and every task index that is a multiple of 64 -- 64x fewer partials for a giant bucket
//    (n/2 points of a selector column: 8192 partials -> 128 leaders).
// 2. k_msm_combine_small / k_msm_combine: sum the leaders of a bucket (one lane / one workgroup).
__device__ __forceinline__ uint32_t bucket_of_task(const uint32_t* __restrict__ toff, uint32_t M, uint32_t v) {
    uint32_t lo_p = 0, hi_p = M;                 // toff[lo_p] <= v < toff[hi_p]
    while (hi_p - lo_p > 1) {
        const uint32_t mid = (lo_p + hi_p) >> 1;
        if (toff[mid] <= v) lo_p = mid; else hi_p = mid;
    }
    return lo_p;
}
// which split buckets k_msm_buckets left to the combination kernels: the heavy ones (positions below H) and, beyond its
// done-counters, those whose tasks straddle a wave boundary
__device__ __forceinline__ bool left_to_kernels(const uint32_t* __restrict__ toff, uint32_t pos, uint32_t H) {
    if (pos < H) return true;
    if (pos < TASK_DONE_MAX) return false;
    const uint32_t t0 = toff[pos], nt = toff[pos + 1] - t0;
    return (t0 & 63u) + nt > 64u;
}
__global__ void __launch_bounds__(256) k_msm_combine_wave(const uint32_t* __restrict__ nmulti, const uint32_t* __restrict__ toff, G1Xyzz29* __restrict__ partial) {
    // A fixed, small grid walks the tasks with a grid stride: a workgroup that finds nothing to do still costs the
    // dispatcher ~15 ns, and a grid sized for the worst case (one lane per possible task: 3000+ workgroups) took 48 us per
    // MSM to establish that a uniform column has NO multi-task bucket at all (profiles/r03_combine_grid.md).
    const uint32_t M = nmulti[0], H = nmulti[2];
    if (H == 0 && nmulti[3] == 0) return;              // every split bucket was put together inside k_msm_buckets: the common case costs one load
    const uint32_t Tm = M ? toff[M] : 0u;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; (v & ~63u) < Tm; v += gridDim.x * blockDim.x) {      // whole waves stay together
        const bool live = v < Tm;
        uint32_t key = 0xFFFFFF00u + lane;           // dead lanes: unique keys, never merged
        G1Xyzz29 acc = identity29();
        if (live) {
            const uint32_t pos = bucket_of_task(toff, M, v);
            if (left_to_kernels(toff, pos, H)) { key = pos; acc = ldg29(partial + v); }     // else: combined inside k_msm_buckets
        }
        const uint32_t prev = __shfl_up(key, 1);
        bool merged = false;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t okey = __shfl_down(key, off);
            const bool take = lane + off < 64u && okey == key;
            if (!__ballot(take)) break;              // segments are contiguous: no pair at this distance, none further
            const G1Xyzz29 other = shfl_down_pt(acc, off);
            if (take) { acc = add29pt(acc, other); merged = true; }
        }
        if (live && merged && (lane == 0 || prev != key)) stg29(partial + v, acc);
    }
}
// leader slots of a bucket whose tasks are [base, base + cnt): base, then the multiples of 64 inside
__device__ __forceinline__ uint32_t leader_count(uint32_t base, uint32_t cnt) { return ((base + cnt - 1) >> 6) - (base >> 6) + 1; }
__device__ __forceinline__ uint32_t leader_slot(uint32_t base, uint32_t i) { return i ? (((base >> 6) + i) << 6) : base; }

constexpr uint32_t COMBINE_SMALL = 32;
// grid of the two grid-stride combination kernels: one lane per task up to a million lanes (a launch that finds nothing to do
// costs 3 us whatever its grid, tools/launch_cost.hip; a column of small values has 262 144 task partials to reduce)
static inline unsigned combine_grid(size_t worst_case_blocks) { return (unsigned)std::min<size_t>(std::max<size_t>(worst_case_blocks, 1), 4096); }
// multi-task buckets with few leaders: one lane each, sequential sum
__global__ void __launch_bounds__(256) k_msm_combine_small(const uint32_t* __restrict__ nmulti, const uint32_t* __restrict__ order, const uint32_t* __restrict__ ntasks,
                                                           const uint32_t* __restrict__ toff, const G1Xyzz29* __restrict__ partial, G1Xyzz29* __restrict__ buckets, uint32_t perm = 0) {
    const uint32_t M = nmulti[0], H = nmulti[2];
    if (H == 0 && nmulti[3] == 0) return;
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {      // small fixed grid, see k_msm_combine_wave
        if (!left_to_kernels(toff, m, H)) continue;                                                      // combined inside k_msm_buckets
        const uint32_t base = toff[m], cnt = leader_count(base, ntasks[m]);
        if (cnt > COMBINE_SMALL) continue;
        G1Xyzz29 acc = ldg29(partial + base);
        for (uint32_t i = 1; i < cnt; ++i) acc = add29pt(acc, ldg29(partial + leader_slot(base, i)));
        stg29(buckets + perm_true(order[m], perm), acc);
    }
}
// one workgroup per multi-task bucket with many leaders: strided sums + LDS tree
__global__ void __launch_bounds__(256) k_msm_combine(const uint32_t* __restrict__ nmulti, const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ ntasks, const uint32_t* __restrict__ toff, const G1Xyzz29* __restrict__ partial,
                                                     G1Xyzz29* __restrict__ buckets, uint32_t perm = 0) {
    __shared__ G1Xyzz29 sh[256];
    const uint32_t total = nmulti[0], H = nmulti[2];
    if (H == 0 && nmulti[3] == 0) return;
    for (uint32_t m = blockIdx.x; m < total; m += gridDim.x) {
        if (!left_to_kernels(toff, m, H)) continue;                                                      // combined inside k_msm_buckets
        const uint32_t p = m, base = toff[p], cnt = leader_count(base, ntasks[p]);
        if (cnt <= COMBINE_SMALL) continue;      // handled by k_msm_combine_small (uniform across the workgroup)
        G1Xyzz29 acc = identity29();
        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) acc = add29pt(acc, ldg29(partial + leader_slot(base, i)));
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off && threadIdx.x + off < cnt) sh[threadIdx.x] = add29pt(sh[threadIdx.x], sh[threadIdx.x + off]);
            __syncthreads();
        }
        if (threadIdx.x == 0) stg29(buckets + perm_true(order[p], perm), sh[0]);
        __syncthreads();
    }
}

// k * P for small k (< 2^16), MSB-first double-and-add
__device__ __forceinline__ G1Xyzz29 mul_small(const G1Xyzz29& p, uint32_t k) {
    G1Xyzz29 acc = identity29();
    for (int bit = 31 - __clz(k | 1); bit >= 0; --bit) {
        acc = dbl29pt(acc);
        if ((k >> bit) & 1) acc = add29pt(acc, p);
    }
    return k ? acc : identity29();
}

constexpr int RED_G_WIDE = 8, RED_G_FOLDED = 2, RED_G_MERGED = 16;
constexpr int RED_THREADS = 256;
// grid: (groups_per_window / RED_THREADS, W); each block writes one partial per (window, block)
// G = buckets folded per lane: 8 keeps the work low when W windows are reduced; 2 keeps the
// dependent chain short when the window tables have already folded everything into one window.
template <int RED_G>
__global__ void __launch_bounds__(RED_THREADS) k_msm_reduce(const G1Xyzz29* __restrict__ buckets, uint32_t B, G1Xyzz29* __restrict__ partial) {
    __shared__ G1Xyzz29 sh[RED_THREADS];
    const uint32_t w = blockIdx.y;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;     // group index in window
    const uint32_t groups = (B + RED_G - 1) / RED_G;
    G1Xyzz29 acc = identity29();
    if (g < groups) {
        const uint32_t b0 = g * RED_G, b1 = min(b0 + RED_G, B);
        G1Xyzz29 running = identity29();
        for (uint32_t b = b1; b-- > b0;) {
            running = add29pt(running, ldg29(buckets + (uint64_t)w * B + b));
            acc = add29pt(acc, running);
        }
        // acc = sum (b - b0 + 1) * B_b ; lift by b0: + b0 * running
        if (b0) acc = add29pt(acc, mul_small(running, b0));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = RED_THREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = add29pt(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(uint64_t)w * gridDim.x + blockIdx.x] = sh[0];
}
// one block per window sums `cnt` partials; output in the canonical R-form XYZZ the host tail reads
__global__ void __launch_bounds__(RED_THREADS) k_msm_window_sum(const G1Xyzz29* __restrict__ partial, uint32_t cnt, G1Xyzz* __restrict__ out_arg,
                                                                const MsmCol* __restrict__ cols = nullptr, uint32_t* ctr = nullptr, uint32_t ctr_step = 0) {
    __shared__ G1Xyzz29 sh[RED_THREADS];
    const uint32_t w = blockIdx.x;
    G1Xyzz* out = cols ? cols[*ctr].out : out_arg;          // graph mode launches one block: every thread reads the counter before thread 0 advances it
    G1Xyzz29 acc = identity29();
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) acc = add29pt(acc, ldg29(partial + (uint64_t)w * cnt + i));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = RED_THREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = add29pt(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stg(out + w, to_std_xyzz(sh[0]));
        if (cols) *ctr += ctr_step;                          // the tree sum above passed several barriers since the counter was read
    }
}

// ---- weighted bucket sum of the merged path: sum_b (b + 1) * B_b over 2^19 .. 2^21 buckets -----------
// With b = g * G + j:  sum_b b B_b = sum_g [sum_j j B_{gG+j}] + G * sum_g g S_g,  S_g = sum_j B_{gG+j}:
// every level turns `count` points into count / G group sums with two short running sums per lane
// (2 G additions, no scalar multiplication) and one partial of the local weighted sums per
// workgroup; the last level (<= 1024 points, one workgroup) lifts its lanes by g * G directly.
//   result = U + A_1 + G (A_2 + G (A_3 + ... G A_last)),   U = sum of all buckets.
// About 2.3 additions per bucket in total, against one ~19-bit scalar multiplication per lane
// before: the reduction of a merged MSM costs a sixth of its accumulation instead of a fifth.
constexpr int WS_G = 8;
constexpr uint32_t WS_LAST_MAX = 1024;
constexpr int WS_MAX_LEVELS = 8;
__device__ __forceinline__ G1Xyzz29 block_sum29(G1Xyzz29 v, G1Xyzz29* sh) {      // 256 threads; result valid in thread 0
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = add29pt(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    const G1Xyzz29 r = sh[0];
    __syncthreads();
    return r;
}
// two sums over the workgroup at once: the additions of a tree level are independent, so the pair costs one
// dependent addition per level, not two
__device__ __forceinline__ void block_sum29x2(G1Xyzz29& a, G1Xyzz29& b, G1Xyzz29* sh) {       // sh: 2 x 256 points
    sh[threadIdx.x] = a;
    sh[256 + threadIdx.x] = b;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const G1Xyzz29 x = add29pt(sh[threadIdx.x], sh[threadIdx.x + off]);
            const G1Xyzz29 y = add29pt(sh[256 + threadIdx.x], sh[256 + threadIdx.x + off]);
            sh[threadIdx.x] = x;
            sh[256 + threadIdx.x] = y;
        }
        __syncthreads();
    }
    a = sh[0];
    b = sh[256];
    __syncthreads();
}
template <bool LAST>
__global__ void __launch_bounds__(256) k_wsum_level(const G1Xyzz29* __restrict__ in, uint32_t count, G1Xyzz29* __restrict__ S_out, G1Xyzz29* __restrict__ Tpart, G1Xyzz29* __restrict__ U_out) {
    __shared__ G1Xyzz29 sh[LAST ? 512 : 256];
    const uint32_t g = blockIdx.x * 256 + threadIdx.x, groups = (count + WS_G - 1) / WS_G;
    G1Xyzz29 acc = identity29(), run = identity29();
    if (g < groups) {
        const uint32_t b0 = g * WS_G, b1 = min(b0 + WS_G, count);
        for (uint32_t b = b1; b-- > b0;) { acc = add29pt(acc, run); run = add29pt(run, ldg29(in + b)); }     // acc = sum_j j * in[b0 + j]
        if (LAST) { if (b0) acc = add29pt(acc, mul_small(run, b0)); }
        else stg29(S_out + g, run);
    }
    if (LAST) {
        block_sum29x2(acc, run, sh);
        if (threadIdx.x == 0) { stg29(Tpart + blockIdx.x, acc); stg29(U_out, run); }
    } else {
        const G1Xyzz29 t = block_sum29(acc, sh);
        if (threadIdx.x == 0) stg29(Tpart + blockIdx.x, t);
    }
}
struct WsumPlan { int levels; uint32_t off[WS_MAX_LEVELS], cnt[WS_MAX_LEVELS]; };      // Tpart ranges per level
// wave w sums the partials of level w (and w + 4) with a shuffle tree, all levels at once; lane 0 of
// wave 0 then folds them:  T = U + A_0 + G (A_1 + G (A_2 + ...))
__global__ void __launch_bounds__(256) k_wsum_final(const G1Xyzz29* __restrict__ Tpart, WsumPlan pl, const G1Xyzz29* __restrict__ U, G1Xyzz* __restrict__ out) {
    __shared__ G1Xyzz29 A[WS_MAX_LEVELS];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (int lv = (int)wave; lv < pl.levels; lv += 4) {
        G1Xyzz29 a = identity29();
        for (uint32_t i = lane; i < pl.cnt[lv]; i += 64) a = add29pt(a, ldg29(Tpart + pl.off[lv] + i));
        for (int off = 32; off > 0; off >>= 1) a = add29pt(a, shfl_down_pt(a, off));
        if (lane == 0) A[lv] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        G1Xyzz29 T = A[pl.levels - 1];
        for (int lv = pl.levels - 2; lv >= 0; --lv) {
            for (int d = 1; d < WS_G; d <<= 1) T = dbl29pt(T);       // * G
            T = add29pt(T, A[lv]);
        }
        stg(out, to_std_xyzz(add29pt(T, ldg29(U))));
    }
}
// enqueue the whole reduction of `buckets[nb]` on stream `st`; scratch: >= nb / 4 + 1024 points
static int wsum_enqueue(zk_ctx* ctx, hipStream_t st, const G1Xyzz29* buckets, uint32_t nb, G1Xyzz29* scratch, G1Xyzz* out) {
    WsumPlan pl;
    pl.levels = 0;
    uint32_t tparts = 0, count = nb;
    // layout of scratch: [U][Tparts ...][S level 1][S level 2]...
    uint32_t t_total = 1;
    for (uint32_t c_ = nb; ; ) {
        const uint32_t groups = (c_ + WS_G - 1) / WS_G;
        t_total += c_ <= WS_LAST_MAX ? 1 : (groups + 255) / 256;
        if (c_ <= WS_LAST_MAX) break;
        c_ = groups;
    }
    G1Xyzz29* U = scratch;
    G1Xyzz29* Tpart = scratch + 1;
    G1Xyzz29* S = scratch + t_total;
    const G1Xyzz29* in = buckets;
    while (count > WS_LAST_MAX) {
        if (pl.levels >= WS_MAX_LEVELS - 1) return ctx->fail(ZK_ERR_UNSUPPORTED, "bucket reduction: too many levels");
        const uint32_t groups = (count + WS_G - 1) / WS_G, blocks = (groups + 255) / 256;
        pl.off[pl.levels] = tparts; pl.cnt[pl.levels] = blocks;
        hipLaunchKernelGGL((k_wsum_level<false>), dim3(blocks), dim3(256), 0, st, in, count, S, Tpart + tparts, (G1Xyzz29*)nullptr);
        tparts += blocks;
        ++pl.levels;
        in = S;
        S += groups;
        count = groups;
    }
    pl.off[pl.levels] = tparts; pl.cnt[pl.levels] = 1;
    hipLaunchKernelGGL((k_wsum_level<true>), dim3(1), dim3(256), 0, st, in, count, (G1Xyzz29*)nullptr, Tpart + tparts, U);
    ++pl.levels;
    hipLaunchKernelGGL(k_wsum_final, dim3(1), dim3(256), 0, st, (const G1Xyzz29*)Tpart, pl, (const G1Xyzz29*)U, out);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}

// bases_rp: device bases already in R' form (SRS cache) or nullptr -> converted into scratch
// d_table (nullable): fixed-base window table for exactly this plan (W windows of tab_stride points)
int msm_batch_tab(zk_ctx* ctx, const Fr* const* d_scalar_ptrs, size_t count, const G1Affine* d_bases, const G1Affine* d_bases_rp, const G1Affine* d_table, size_t tab_stride,
                  size_t n, G1Affine* h_out, MsmStageFn stage, void* stage_user) {
    if (count == 0) return ZK_OK;
    if (n == 0) { memset(h_out, 0, sizeof(G1Affine) * count); return ZK_OK; }
    if (n >= (1ull << 31)) return ctx->fail(ZK_ERR_UNSUPPORTED, "MSM larger than 2^31-1 points");
    const MsmPlan pl = make_plan(n);
    const uint32_t nb = (uint32_t)pl.W * pl.B;
    // bucket offsets and task cursors are 32-bit: n * W (point, window) entries must stay below 2^32 (n <= 2^27 at 16 windows)
    if ((uint64_t)n * pl.W >= (1ull << 32)) return ctx->fail(ZK_ERR_UNSUPPORTED, "MSM of %zu points x %d windows exceeds 2^32 index entries: split it", n, pl.W);

    // u32 workspace: slice_counts[4 nb] | slice_off[4 nb + 4] (both 16-B aligned) | counts[nb] | size_hist[256] nmulti[4] wflag[64] |
    //                offsets[nb+1] | order[nb] | ntasks[nb] | toff[nb+1] | block_tot[scan_blocks_s + scan_blocks] | idx[n*W] |
    //                (16-B aligned) dig[W*n_pad u16]
    const uint32_t scan_blocks = (nb + SCAN_T * SCAN_ITEMS - 1) / (SCAN_T * SCAN_ITEMS);
    const uint32_t scan_blocks_s = (nb + SCAN_T - 1) / SCAN_T;                    // scan over the [bucket][slice] counters: one bucket per thread
    const uint64_t n_pad = ((uint64_t)n + 16 * MSM_SLICES - 1) & ~(uint64_t)(16 * MSM_SLICES - 1);
    const size_t dig_words = (size_t)(n_pad * pl.W + 1) / 2 + 4;
    const size_t head_words = (size_t)nb * (2 * MSM_SLICES + 5) + 4 + 2 + SIZE_BINS + 4 + MSM_WFLAGS + TASK_DONE_MAX + (size_t)scan_blocks_s + scan_blocks + (size_t)n * pl.W;
    const size_t words = head_words + 4 + dig_words;
    uint32_t* ws = (uint32_t*)ctx->get_scratch(SC_MSM_KEYS, words * 4);
    if (!ws) return ZK_ERR_OOM;
    uint32_t* slice_counts = ws;
    uint32_t* slice_off = slice_counts + (size_t)nb * MSM_SLICES;
    uint32_t* counts = slice_off + (size_t)nb * MSM_SLICES + 4;
    uint32_t* size_hist = counts + nb;
    uint32_t* nmulti = size_hist + SIZE_BINS;
    uint32_t* wflag = nmulti + 4;          // MSM_WFLAGS words: W <= 64 windows (c >= 4)
    uint32_t* offsets = wflag + MSM_WFLAGS + TASK_DONE_MAX;
    uint32_t* order = offsets + nb + 1;
    uint32_t* ntasks = order + nb;
    uint32_t* toff = ntasks + nb;
    uint32_t* block_tot = toff + nb + 1;
    uint32_t* block_tot2 = block_tot + scan_blocks_s;
    uint32_t* idx = block_tot2 + scan_blocks;
    uint16_t* dig = reinterpret_cast<uint16_t*>(ws + ((head_words + 3) & ~(size_t)3));   // 16-B aligned
    int range_bits = pl.c - 1;
    if (range_bits > MSM_RANGE_MAX_BITS) range_bits = MSM_RANGE_MAX_BITS;
    const dim3 sweep_grid(8u * ((pl.W + 7) / 8) * (pl.B >> range_bits) * MSM_SLICES);
    // a lone MSM is latency-bound (short chains: G = 2); in a batch the reduction hides under the
    // next MSM and only its work counts (G = 8)
    const bool short_chain = d_table && count == 1;
    const int red_g = short_chain ? RED_G_FOLDED : RED_G_WIDE;
    const uint32_t red_blocks = ((pl.B + red_g - 1) / red_g + RED_THREADS - 1) / RED_THREADS;
    const size_t max_tasks = (size_t)nb + std::max(((size_t)n * pl.W) / TASK_CAP, (size_t)TASK_TARGET) + 1;   // tasks <= points / task size + buckets
    const size_t npts29 = (size_t)nb + (size_t)pl.W * red_blocks + max_tasks + pl.B;
    if (!d_table) tab_stride = 0; else d_bases_rp = d_table;
    const int red_W = d_table ? 1 : pl.W;     // windows the weighted reduction has to handle
    // bucket state is double-buffered: the reduction of MSM i (side stream) overlaps phase 1 of MSM i+1
    char* bkbuf[2];
    bkbuf[0] = (char*)ctx->get_scratch(SC_MSM_BUCKETS, sizeof(G1Xyzz29) * npts29);
    bkbuf[1] = count > 1 ? (char*)ctx->get_scratch(SC_MSM_BUCKETS2, sizeof(G1Xyzz29) * npts29) : bkbuf[0];
    // window sums of every MSM of the batch, then one private copy of the window flags per MSM (the
    // fold on the side stream reads them while the main stream already recodes the next column)
    G1Xyzz* wsum_all = (G1Xyzz*)ctx->get_scratch(SC_MSM_RESULTS, sizeof(G1Xyzz) * pl.W * count + 4 * MSM_WFLAGS * count);
    if (!bkbuf[0] || !bkbuf[1] || !wsum_all) return ZK_ERR_OOM;
    if (!ctx->stream2) ZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    if (!ctx->ev_p1[0])
        for (int i = 0; i < 3; ++i) {
            ZK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_p1[i], hipEventDisableTiming));
            ZK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_p2[i], hipEventDisableTiming));
        }

    const dim3 gs((unsigned)((n + 255) / 256)), ts(256);
    if (!d_bases_rp) {
        G1Affine* conv = (G1Affine*)ctx->get_scratch(SC_MSM_MISC, sizeof(G1Affine) * n);
        if (!conv) return ZK_ERR_OOM;
        ZkProfScope ps(ctx, "msm_bases_convert");
        hipLaunchKernelGGL(k_bases_to_rprime, gs, ts, 0, ctx->stream, d_bases, conv, (uint64_t)n);
        ZK_CHECK_LAUNCH(ctx);
        d_bases_rp = conv;
    }
    if (stage) { int rc = stage(stage_user, 0); if (rc) return rc; }
  for (size_t it = 0; it < count; ++it) {
    const int par = (int)(it & 1);
    const Fr* d_scalars = d_scalar_ptrs[it];
    G1Xyzz29* buckets = (G1Xyzz29*)bkbuf[par];
    G1Xyzz29* partial = buckets + nb;
    G1Xyzz29* task_partial = partial + (size_t)pl.W * red_blocks;
    G1Xyzz29* folded = task_partial + max_tasks;
    G1Xyzz* wsum = wsum_all + it * pl.W;
    uint32_t* wflag_it = reinterpret_cast<uint32_t*>(wsum_all + (size_t)pl.W * count) + it * MSM_WFLAGS;
    if (it >= 2) ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_p2[par], 0));   // reduce(it-2) must be done with this buffer
    {
        ZkProfScope ps(ctx, "msm_sort");
        ZK_HIP(ctx, hipMemsetAsync(size_hist, 0, (size_t)(SIZE_BINS + 4 + MSM_WFLAGS + TASK_DONE_MAX) * 4, ctx->stream));   // size_hist + nmulti + wflag
        launch_digits(pl.c, dim3((unsigned)((n_pad / 2 + 255) / 256)), ctx->stream, d_scalars, (uint64_t)n, n_pad, dig, wflag);
        hipLaunchKernelGGL((k_msm_lds_sweep<false>), sweep_grid, dim3(1024), 0, ctx->stream, (const uint16_t*)dig, n_pad, range_bits, pl.B, slice_counts, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)wflag, (uint32_t)pl.W);
        ZK_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL(k_scan_u32_a, dim3(scan_blocks_s), dim3(SCAN_T), 0, ctx->stream, (const uint32_t*)slice_counts, nb * MSM_SLICES, slice_off, block_tot);
        hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, ctx->stream, block_tot, scan_blocks_s, slice_off, nb * MSM_SLICES, offsets + nb);
        hipLaunchKernelGGL(k_scan_u32_c, dim3(scan_blocks_s), dim3(SCAN_T), 0, ctx->stream, (const uint32_t*)slice_counts, nb, slice_off, (const uint32_t*)block_tot, offsets, counts, size_hist);
        hipLaunchKernelGGL(k_size_bins_scan, dim3(1), dim3(64), 0, ctx->stream, size_hist, (const uint32_t*)(offsets + nb));
        hipLaunchKernelGGL(k_order_buckets, dim3(scan_blocks), dim3(SCAN_T), 0, ctx->stream, (const uint32_t*)counts, nb, size_hist, order, ntasks);
        hipLaunchKernelGGL(k_scan_u32_a, dim3(scan_blocks), dim3(SCAN_T), 0, ctx->stream, (const uint32_t*)ntasks, nb, toff, block_tot2);
        hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, ctx->stream, block_tot2, scan_blocks, toff, nb, (uint32_t*)nullptr);
        hipLaunchKernelGGL(k_task_offsets, dim3(scan_blocks), dim3(SCAN_T), 0, ctx->stream, nb, toff, (const uint32_t*)block_tot2);
        ZK_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL((k_msm_lds_sweep<true>), sweep_grid, dim3(1024), 0, ctx->stream, (const uint16_t*)dig, n_pad, range_bits, pl.B, (uint32_t*)nullptr, (const uint32_t*)slice_off, idx, (const uint32_t*)wflag, (uint32_t)pl.W);
        ZK_CHECK_LAUNCH(ctx);
    }
    ZK_HIP(ctx, hipMemcpyAsync(wflag_it, wflag, MSM_WFLAGS * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    {
        ZkProfScope ps(ctx, "msm_buckets");
        // multi-task buckets first (they are the long poles), then one lane per ordinary bucket
        hipLaunchKernelGGL(k_msm_buckets, dim3((unsigned)((max_tasks + 255) / 256)), dim3(256), 0, ctx->stream, d_bases_rp, (const uint32_t*)offsets, (const uint32_t*)idx,
                           (const uint32_t*)order, (const uint32_t*)toff, (const uint32_t*)nmulti, nb, buckets, task_partial, pl.c - 1, (uint64_t)tab_stride, (const uint32_t*)wflag);
    }
    {
        ZkProfScope ps(ctx, "msm_combine");
        hipLaunchKernelGGL(k_msm_combine_wave, dim3(combine_grid((max_tasks + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)nmulti, (const uint32_t*)toff, task_partial);
        hipLaunchKernelGGL(k_msm_combine_small, dim3(combine_grid((nb + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)nmulti, (const uint32_t*)order,
                           (const uint32_t*)ntasks, (const uint32_t*)toff, (const G1Xyzz29*)task_partial, buckets);
        hipLaunchKernelGGL(k_msm_combine, dim3(256), dim3(256), 0, ctx->stream, (const uint32_t*)nmulti, (const uint32_t*)order,
                           (const uint32_t*)ntasks, (const uint32_t*)toff, (const G1Xyzz29*)task_partial, buckets);
        ZK_CHECK_LAUNCH(ctx);
    }
    ZK_HIP(ctx, hipEventRecord(ctx->ev_p1[par], ctx->stream));
    ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_p1[par], 0));
    {   // latency-bound tail on the side stream: it hides under the next MSM's phase 1
        ZkProfScope ps(ctx, "msm_reduce", ctx->stream2);
        const G1Xyzz29* red_in = buckets;
        if (d_table) {
            hipLaunchKernelGGL(k_msm_fold_windows, dim3((pl.B + 63) / 64), dim3(256), 0, ctx->stream2, (const G1Xyzz29*)buckets, pl.B, pl.W, folded, (const uint32_t*)wflag_it);
            red_in = folded;
        }
        if (short_chain) hipLaunchKernelGGL((k_msm_reduce<RED_G_FOLDED>), dim3(red_blocks, red_W), dim3(RED_THREADS), 0, ctx->stream2, red_in, pl.B, partial);
        else hipLaunchKernelGGL((k_msm_reduce<RED_G_WIDE>), dim3(red_blocks, red_W), dim3(RED_THREADS), 0, ctx->stream2, red_in, pl.B, partial);
        ZK_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL(k_msm_window_sum, dim3(red_W), dim3(RED_THREADS), 0, ctx->stream2, (const G1Xyzz29*)partial, red_blocks, wsum);
        ZK_CHECK_LAUNCH(ctx);
    }
    ZK_HIP(ctx, hipEventRecord(ctx->ev_p2[par], ctx->stream2));
    if (stage && it + 1 < count) { int rc = stage(stage_user, it + 1); if (rc) return rc; }
  }
    // join: the main stream waits for the outstanding reductions, then one copy of all window sums
    ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_p2[0], 0));
    if (count > 1) ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_p2[1], 0));
    std::vector<G1Xyzz> hw((size_t)pl.W * count);
    ZK_HIP(ctx, hipMemcpyAsync(hw.data(), wsum_all, sizeof(G1Xyzz) * pl.W * count, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t it = 0; it < count; ++it) host::msm_tail(hw.data() + it * pl.W, red_W, pl.c, h_out + it);
    return ZK_OK;
}
#define PK_TRY_MSM(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)
// Blinded tails.  A witness column of small values ends in a few rows of field-sized blinding values (halo2: the last
// blinding_factors + 1 rows); inside the main MSM they would put a handful of entries into EVERY window and undo what the
// per-window path gains by skipping empty ones (2^18 bit columns: 0.34 instead of 0.21 ms).  They are committed here instead:
// one workgroup per column, lane (r, w) lifts the unsigned c-bit digit w of tail scalar r over the merged plan's table entry
// 2^(c w) P_{n_main + r} by double-and-add, the workgroup sums its lanes.  flags[col] != 1: identity.
__global__ void __launch_bounds__(1024) k_msm_tails(const Fr* const* __restrict__ scalar_ptrs, const uint8_t* __restrict__ flags, uint64_t n_main, uint32_t tail,
                                                    const G1Affine* __restrict__ table, uint64_t tab_stride, int c, int W, int top_shift, G1Xyzz* __restrict__ out) {
    __shared__ G1Xyzz29 wsum[16];
    const uint32_t col = blockIdx.x;
    G1Xyzz29 acc = identity29();
    if (flags[col] == 1) {
        const Fr* scalars = scalar_ptrs[col];
        for (uint32_t pair = threadIdx.x; pair < tail * (uint32_t)W; pair += blockDim.x) {
            const uint32_t r = pair / (uint32_t)W, w = pair % (uint32_t)W;
            const Fr sc = from_mont(ldg(scalars + n_main + r));
            const int bit = (int)w * c, limb = bit >> 5, sh = bit & 31;
            uint64_t d = limb < 8 ? ((uint64_t)sc.l[limb] >> sh) : 0ull;
            if (limb + 1 < 8) d |= (uint64_t)sc.l[limb + 1] << (32 - sh);
            d &= (1ull << c) - 1;
            int bits = c;
            if ((int)w == W - 1) { d <<= top_shift; bits = c + top_shift; }      // the table entry of the top window is 2^top_shift short (MsmPlan::top_shift)
            if (d == 0) continue;
            const G1Affine29 base = load_affine29(table + (uint64_t)w * tab_stride + n_main + r);
            G1Xyzz29 t = identity29();
            for (int b = bits - 1; b >= 0; --b) {
                t = dbl29pt(t);
                if ((d >> b) & 1) t = madd29(t, base);
            }
            acc = add29pt(acc, t);
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc = add29pt(acc, shfl_down_pt(acc, off));
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        G1Xyzz29 s = wsum[0];
        for (uint32_t i = 1; i < blockDim.x / 64; ++i) s = add29pt(s, wsum[i]);
        stg(out + col, to_std_xyzz(s));
    }
}
// ---- merged-window MSM over an SRS window table ----------------------------------------------------
static MsmPlan make_plan_merged(uint32_t k_srs) {
    int c = (int)k_srs;
    if (c < MSM_M_MIN_C) c = MSM_M_MIN_C;
    if (c > MSM_M_MAX_C) c = MSM_M_MAX_C;
    if (const char* e = getenv("ZK_MSM_C")) { const int v = atoi(e); if (v >= MSM_M_MIN_C && v <= MSM_M_MAX_C) c = v; }   // measurement knob
    MsmPlan p;
    p.c = c;
    p.W = (256 + c - 1) / c;
    p.B = 1u << (c - 1);
    // largest shift that keeps the top digit (leading bits of a canonical scalar, plus the carry of the window below)
    // at or below 2^(c-1)
    const int low = c * (p.W - 1);
    uint64_t top_max = 1;
    if (low < 254) {
        uint64_t v = 0;      // (modulus - 1) >> low, from the 8 x 32-bit limbs (the modulus is odd: -1 only clears bit 0)
        for (int b = 0; b < 40 && low + b < 256; ++b) {
            const int bit = low + b;
            const uint32_t limb = bit < 32 ? FrP::M(0) - 1u : FrP::M(bit >> 5);
            v |= (uint64_t)((limb >> (bit & 31)) & 1u) << b;
        }
        top_max = v + 1;
    }
    p.top_shift = 0;
    while (p.top_shift + 1 <= c - 1 && (top_max << (p.top_shift + 1)) <= (1ull << (c - 1))) ++p.top_shift;
    if (const char* e = getenv("ZK_MSM_TOP_SHIFT")) { const int v = atoi(e); if (v >= 0 && v <= p.top_shift) p.top_shift = v; }   // measurement knob
    return p;
}
template <bool SCATTER>
static void launch_partition(int c, dim3 grid, hipStream_t st, const Fr* scalars, uint64_t n, int range_bits, uint32_t* hist, const uint32_t* hist_off, uint64_t* entries, uint64_t tab_stride, int top_shift,
                             const MsmCol* cols = nullptr, const uint32_t* ctr = nullptr) {
#define ZK_PART_CASE(C) case C: hipLaunchKernelGGL((k_msm_m_partition<C, SCATTER>), grid, dim3(256), 0, st, scalars, n, range_bits, hist, hist_off, entries, tab_stride, top_shift, cols, ctr); break;
    switch (c) {
        ZK_PART_CASE(8) ZK_PART_CASE(9) ZK_PART_CASE(10) ZK_PART_CASE(11) ZK_PART_CASE(12) ZK_PART_CASE(13) ZK_PART_CASE(14) ZK_PART_CASE(15)
        ZK_PART_CASE(16) ZK_PART_CASE(17) ZK_PART_CASE(18) ZK_PART_CASE(19) ZK_PART_CASE(20) ZK_PART_CASE(21) ZK_PART_CASE(22)
    }
#undef ZK_PART_CASE
}
template <bool SCATTER>
static void launch_gm_partition(int c, dim3 grid, hipStream_t st, const GmCols& cols, uint64_t n, int range_bits, uint32_t* hist, const uint32_t* hist_off, uint64_t* entries, uint64_t tab_stride,
                                uint32_t bins_per_col, uint32_t set_mask) {
#define ZK_GMP_CASE(C) case C: hipLaunchKernelGGL((k_msm_gm_partition<C, SCATTER>), grid, dim3(256), 0, st, cols, n, range_bits, hist, hist_off, entries, tab_stride, bins_per_col, set_mask); break;
    switch (c) {
        ZK_GMP_CASE(8) ZK_GMP_CASE(9) ZK_GMP_CASE(10) ZK_GMP_CASE(11) ZK_GMP_CASE(12) ZK_GMP_CASE(13) ZK_GMP_CASE(14) ZK_GMP_CASE(15) ZK_GMP_CASE(16)
    }
#undef ZK_GMP_CASE
}
static int launch_scatter_staged(zk_ctx* ctx, int c, int W, dim3 grid, const Fr* scalars, uint64_t n, int range_bits, const uint32_t* hist_off, uint64_t* entries, uint64_t tab_stride, int top_shift,
                                 const MsmCol* cols = nullptr, const uint32_t* ctr = nullptr, hipStream_t st = nullptr) {
    const size_t lds = scatter_staged_lds(W);
    if (!st) st = ctx->stream;
#define ZK_SS_CASE(C) case C: \
        if (!(ctx->msm_attr_set & (1u << C))) { ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_msm_m_scatter_staged<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ctx->msm_attr_set |= 1u << C; } \
        hipLaunchKernelGGL((k_msm_m_scatter_staged<C>), grid, dim3(1024), lds, st, scalars, n, range_bits, hist_off, entries, tab_stride, top_shift, cols, ctr); break;
    switch (c) { ZK_SS_CASE(19) ZK_SS_CASE(20) ZK_SS_CASE(21) ZK_SS_CASE(22) default: return ctx->fail(ZK_ERR_UNSUPPORTED, "staged scatter: window size %d", c); }
#undef ZK_SS_CASE
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}
// d_table: [W][tab_stride] affine points, table[w][i] = 2^(c w) * P_i in R' form, built for plan `pl`.
// Same pipelining as msm_batch_tab: the reduction of MSM i runs on the side stream under MSM i + 1.
// narrow[it] != 0 (with d_table_n, the per-window table of plan pl_n): column `it` is expected to fill
// only a few windows (witness columns of small values, selectors, lookup multiplicities) and takes the
// per-window path -- bucket sets of empty windows are never touched there, whereas the merged path
// always pays for its 2^(c-1) shared buckets.  The two paths alternate freely inside one pipelined batch.
static thread_local uint32_t tl_gm_group_cap = 0;       // upper bound on the columns of a GM group while a batch is retried with smaller groups (0 = none)
int msm_batch_merged(zk_ctx* ctx, const Fr* const* d_scalar_ptrs, size_t count, const G1Affine* d_table, size_t tab_stride, const MsmPlan& pl,
                     size_t n, G1Affine* h_out, MsmStageFn stage, void* stage_user, const G1Affine* d_table_n = nullptr, const MsmPlan* pl_n = nullptr, const uint8_t* narrow = nullptr) {
    if (count == 0) return ZK_OK;
    if (n == 0) { memset(h_out, 0, sizeof(G1Affine) * count); return ZK_OK; }
    const auto t_batch0 = std::chrono::steady_clock::now();
    bool any_narrow = false;
    if (d_table_n && pl_n && narrow) for (size_t i = 0; i < count; ++i) any_narrow |= narrow[i] == 1;
    // the per-window path indexes n * W entries with 32 bits; decided BEFORE the blinded tails are split off: a column that
    // takes the merged path over all n rows must not have its tail committed a second time by k_msm_tails
    if (any_narrow && (uint64_t)n * pl_n->W >= (1ull << 32)) any_narrow = false;
    // blinded tails of the hint-1 columns (zk_ctx::msm_blinded_tail): committed by k_msm_tails, the main MSM stops in front of them
    const uint32_t tail_rows = (any_narrow && ctx->msm_blinded_tail && (size_t)ctx->msm_blinded_tail + 1024 <= n && ctx->msm_blinded_tail <= 256) ? ctx->msm_blinded_tail : 0u;
    const uint64_t n_narrow = (uint64_t)n - tail_rows;
    // ---- per-window ("narrow") path: sizes and workspace, as in msm_batch_tab with a window table
    const MsmPlan pn = any_narrow ? *pl_n : MsmPlan{4, 64, 8};
    // Consecutive small-valued columns share ONE launch sequence: column j of a group owns windows [j W, (j + 1) W) of a
    // (group x W)-window MSM over the same per-window table -- digits, LDS sweeps, scans, task split, accumulation and
    // combination run once per group, the fold / reduction with the column on blockIdx.y.  Such a column is a chain of sixteen
    // small, latency-bound launches (a few hundred thousand entries); a group amortises every launch over up to eight columns
    // (MSM_WFLAGS window flags).  ZK_MSM_NARROW_GROUP=1 turns it off, 2 / 4 / 8 set the group size (measurement knob; 2^20 rows,
    // 30-bit values: 0.68 / 0.51 / 0.43 / 0.38 ms per column, tools/gpu_r3u.sh).
    uint32_t NG = 1;
    if (any_narrow) {
        NG = std::min<uint32_t>((uint32_t)GM_MAX_COLS, MSM_WFLAGS / (uint32_t)pn.W);       // sixteen at c = 16: a group's dozen short launches (scans, size bins, task split) cost the same for eight
                                                                                            // columns or sixteen -- 60/30/10 columns 0.265 -> 0.255 ms each in a batch, the sort 0.81 ms per eight -> 1.45 per sixteen
        if (const char* e = getenv("ZK_MSM_NARROW_GROUP")) { const int v = atoi(e); if (v >= 1 && v <= GM_MAX_COLS) NG = std::min<uint32_t>((uint32_t)v, MSM_WFLAGS / (uint32_t)pn.W); }
        if (const char* e = getenv("ZK_MSM_SORT_AHEAD")) if (atoi(e) == 1) NG = 1;
        if (tl_gm_group_cap && NG > tl_gm_group_cap) NG = tl_gm_group_cap;            // a retry after the group's sort workspace did not fit (below)
        while (NG > 1 && (uint64_t)n * pn.W * NG >= (1ull << 32)) --NG;
        if (NG < 1) NG = 1;
    }
    const uint32_t Wg = (uint32_t)pn.W * NG;
    const uint32_t nbN = Wg * pn.B;
    const uint32_t scan_blocks_N = (nbN + SCAN_T * SCAN_ITEMS - 1) / (SCAN_T * SCAN_ITEMS), scan_blocks_sN = (nbN + SCAN_T - 1) / SCAN_T;
    const uint64_t n_pad = ((uint64_t)n + 16 * MSM_SLICES - 1) & ~(uint64_t)(16 * MSM_SLICES - 1);
    const size_t dig_words = (size_t)(n_pad * Wg + 1) / 2 + 4;
    const size_t head_words_N = (size_t)nbN * (2 * MSM_SLICES + 5) + 4 + 2 + SIZE_BINS + 4 + MSM_WFLAGS + TASK_DONE_MAX + (size_t)scan_blocks_sN + scan_blocks_N + (size_t)n * Wg;
    // Groups of small-valued columns, sorted like the merged path ("GM"; ZK_MSM_NARROW_GM=0 keeps the digit matrix + LDS sweeps):
    // the digit matrix of a group is 16 x n codes per column of which a witness-like column fills a tenth, and every row of it
    // was streamed by sixteen range workgroups, twice -- 0.22 ms of sort per column for 0.12 ms of accumulation.  Here the non-zero
    // digits become entries keyed by (column, bucket) -- the per-window table gives every window's bucket b the same weight, so a
    // column needs ONE set of 2^(c-1) buckets, not one per window -- and go through the merged path's partition / one-launch
    // counting sort: column j of the group owns partitions [j * bpc, (j + 1) * bpc) and buckets [j B, (j + 1) B); the fold of the
    // windows disappears (the accumulation already sums them), reduction and window sum run per column as before.
    const bool gm = any_narrow && pn.c >= MSM_M_MIN_C && pn.c <= 16 && NG <= (uint32_t)GM_MAX_COLS && !(getenv("ZK_MSM_NARROW_GM") && atoi(getenv("ZK_MSM_NARROW_GM")) == 0);
    // A column's windows are dealt over SG bucket sets (window w -> set w mod SG): a witness-like column (10 % field-sized cells) puts
    // ~2 M entries into its buckets -- 61 per bucket with one set of 2^15, every bucket split into two unequal tasks and put together
    // again afterwards (accumulation + combination 2.2 ms per group of eight against the 1.0 ms the additions cost); with two sets
    // the buckets hold 26-35 entries, one task each.  ZK_MSM_GM_SETS = 1 / 2 / 4 (measurement knob).
    uint32_t SG = 2;
    if (const char* e = getenv("ZK_MSM_GM_SETS")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) SG = (uint32_t)v; }
    int range_bits_G = pn.c - 1 - 7;                                  // 128 * SG partitions per column of 2^(c-8) buckets each
    if (range_bits_G < 0) range_bits_G = 0;
    const uint32_t bpcG = (SG * pn.B) >> range_bits_G;                // partitions per column
    const uint32_t perm_G = ((uint32_t)range_bits_G << 8) | (uint32_t)(pn.c - 1 - range_bits_G);      // the GM sort numbers buckets low bits first (perm_true)
    const uint32_t nwgG = (uint32_t)((n_narrow + MSM_M_CHUNK - 1) / MSM_M_CHUNK);      // partition workgroups per column
    const uint32_t nbG = NG * SG * pn.B;
    const uint64_t entG = (uint64_t)n * pn.W * NG;                    // worst case: every digit of every column non-zero
    const uint32_t hist_cnt_G = NG * bpcG * nwgG;
    const uint32_t scan_blocks_G = (nbG + SCAN_T * SCAN_ITEMS - 1) / (SCAN_T * SCAN_ITEMS), scan_blocks_hG = (hist_cnt_G + SCAN_T * SCAN_ITEMS - 1) / (SCAN_T * SCAN_ITEMS);
    // u32 workspace of the GM sort: counts[nbG] | size_hist nmulti wflags done | offsets[nbG + 1] | order[nbG] | ntasks[nbG] | toff[nbG + 1] | block_tot2 | block_tot3 |
    //                               hist[hist_cnt_G] | hist_off[hist_cnt_G + 1] | idx[entG] | (8-B aligned) entries[entG] u64
    const size_t head_words_G = (size_t)nbG * 5 + 2 + SIZE_BINS + 4 + MSM_WFLAGS + TASK_DONE_MAX + scan_blocks_G + scan_blocks_hG + 2 * (size_t)hist_cnt_G + 2 + (size_t)entG;
    const size_t words_G = gm ? head_words_G + 4 + 2 * (size_t)entG : 0;
    const size_t words_N = any_narrow ? std::max(head_words_N + 4 + dig_words, words_G) : 0;
    int range_bits_N = pn.c - 1;
    if (range_bits_N > MSM_RANGE_MAX_BITS) range_bits_N = MSM_RANGE_MAX_BITS;
    const uint32_t red_blocks_N = ((pn.B + RED_G_WIDE - 1) / RED_G_WIDE + RED_THREADS - 1) / RED_THREADS;
    const size_t max_tasks_N = (size_t)nbN + std::max(((size_t)n * Wg) / TASK_CAP, (size_t)TASK_TARGET) + 1;
    const size_t npts29_N = any_narrow ? (size_t)nbN + (size_t)red_blocks_N * NG * 4 + max_tasks_N + (size_t)pn.B * NG : 0;      // x 4: the GM path reduces up to four bucket sets per column
    const uint32_t nb = pl.B;
    const uint64_t max_entries = (uint64_t)n * pl.W;
    // table indices carry the sign in bit 31, cursors are 32-bit
    if ((uint64_t)pl.W * tab_stride >= (1ull << 31) || max_entries >= (1ull << 32)) return ctx->fail(ZK_ERR_UNSUPPORTED, "MSM of %zu points x %d windows exceeds the 32-bit entry space: split it", n, pl.W);
    int range_bits = pl.c - 1;
    if (range_bits > MSM_RANGE_MAX_BITS) range_bits = MSM_RANGE_MAX_BITS;
    // one-launch partition sort (k_msm_m_binsort) when 1024 partitions are small enough for its LDS staging buffer;
    // ZK_MSM_BINSORT=0 keeps the sliced count / scan / scatter sequence (measurement knob)
    const char* env_bs = getenv("ZK_MSM_BINSORT");
    bool binsort = !(env_bs && atoi(env_bs) == 0);
    if (binsort) {
        int rb = pl.c - 1 - 10;
        if (rb < 0) rb = 0;
        const uint64_t per_partition = max_entries / (nb >> rb);
        if (per_partition + per_partition / 8 <= MSM_M_STAGE) range_bits = rb;
        else binsort = false;
    }
    const uint32_t nbins = nb >> range_bits;
    uint32_t chunk = MSM_M_CHUNK;
    if (const char* e = getenv("ZK_MSM_CHUNK")) { const int v = atoi(e); if (v >= 256 && v <= 65536) chunk = (uint32_t)v; }   // measurement knob
    // scatter with LDS-staged runs (k_msm_m_scatter_staged): window sizes whose 1024 x W entries fit the LDS, and only next to the
    // one-launch partition sort (same 1024-partition layout); ZK_MSM_STAGED=0 keeps the direct scatter (measurement knob)
    const char* env_ss = getenv("ZK_MSM_STAGED");
    const bool staged_scatter = binsort && pl.c >= 19 && scatter_staged_lds(pl.W) <= (size_t)150 * 1024 && !(env_ss && atoi(env_ss) == 0);
    if (staged_scatter) chunk = MSM_M_SCHUNK;
    const uint32_t nwg = (uint32_t)((n + chunk - 1) / chunk);
    const uint32_t hist_cnt = nbins * nwg;
    const uint32_t scan_blocks = (nb + SCAN_T * SCAN_ITEMS - 1) / (SCAN_T * SCAN_ITEMS);
    const uint32_t scan_blocks_s = (nb + SCAN_T - 1) / SCAN_T;
    const uint32_t scan_blocks_h = (hist_cnt + SCAN_T * SCAN_ITEMS - 1) / (SCAN_T * SCAN_ITEMS);
    // u32 workspace: slice_counts[4 nb] | slice_off[4 nb + 4] | counts[nb] | size_hist[256] nmulti[4] pad[64] | offsets[nb+1] | order[nb] |
    //                ntasks[nb] | toff[nb+1] | block_tot[...] | hist[hist_cnt] | hist_off[hist_cnt + 1] | idx[n W] | (8-B aligned) entries[n W] u64
    const size_t head_words = (size_t)nb * (2 * MSM_SLICES + 5) + 4 + 2 + SIZE_BINS + 4 + MSM_WFLAGS + TASK_DONE_MAX + (size_t)scan_blocks_s + scan_blocks + scan_blocks_h + 2 * (size_t)hist_cnt + 2 + max_entries;
    const size_t words = (std::max(head_words + 4 + 2 * max_entries, words_N) + 63) & ~(size_t)63;
    // Below 2^20 points an MSM does not fill the device: its sort / accumulation / combination is a chain of some twenty
    // short launches.  Two such chains then run side by side -- even columns on the context's stream, odd columns on the
    // third side stream, each with its own sort workspace -- and hide each other's latency.  ZK_MSM_PIPES overrides.
    int npipe = (n <= ((size_t)1 << 19) && count >= 2) ? 2 : 1;
    if (const char* e = getenv("ZK_MSM_PIPES")) { const int v = atoi(e); if (v == 1 || (v == 2 && count >= 2)) npipe = v; }
    // Graph mode (below): the launch sequence of a column is captured once per pipeline and replayed -- at these sizes the
    // host's launch rate, ~30 API calls per column, is what bounds a batch.  ZK_MSM_GRAPH=0 disables it.
    const char* env_graph = getenv("ZK_MSM_GRAPH");
    const bool want_graph = n <= ((size_t)1 << 19) && n >= 1024 && count >= 4 && !ctx->prof_on && !ctx->msm_graph_broken && !(env_graph && atoi(env_graph) == 0) && !(any_narrow && NG > 1);
    constexpr int GP = 4;                     // pipelines of the graph mode: the context's stream and the three side streams
    const char* env_sa = getenv("ZK_MSM_SORT_AHEAD");
    const int ws_copies = want_graph ? GP : ((npipe == 2 || (count >= 2 && env_sa && atoi(env_sa) == 1)) ? 2 : 1);
    uint32_t* ws = (uint32_t*)ctx->get_scratch(SC_MSM_KEYS, words * 4 * ws_copies);
    if (!ws && any_narrow && NG > 1 && words_N > head_words + 4 + 2 * max_entries) {
        // the group sort's workspace is sized for the worst case (every digit of all NG columns non-zero: 3 GiB per copy at 2^20 rows and
        // sixteen columns): with memory short, smaller groups are tried before the batch is given up
        struct Cap { uint32_t prev; explicit Cap(uint32_t v) : prev(tl_gm_group_cap) { tl_gm_group_cap = v; } ~Cap() { tl_gm_group_cap = prev; } } cap(NG / 2);
        (void)hipGetLastError();
        return msm_batch_merged(ctx, d_scalar_ptrs, count, d_table, tab_stride, pl, n, h_out, stage, stage_user, d_table_n, pl_n, narrow);
    }
    if (!ws) return ZK_ERR_OOM;
    uint32_t* slice_counts = ws;
    uint32_t* slice_off = slice_counts + (size_t)nb * MSM_SLICES;
    uint32_t* counts = slice_off + (size_t)nb * MSM_SLICES + 4;
    uint32_t* size_hist = counts + nb;
    uint32_t* nmulti = size_hist + SIZE_BINS;
    uint32_t* offsets = nmulti + 4 + MSM_WFLAGS + TASK_DONE_MAX;
    uint32_t* order = offsets + nb + 1;
    uint32_t* ntasks = order + nb;
    uint32_t* toff = ntasks + nb;
    uint32_t* block_tot = toff + nb + 1;
    uint32_t* block_tot2 = block_tot + scan_blocks_s;
    uint32_t* block_tot3 = block_tot2 + scan_blocks;
    uint32_t* hist = block_tot3 + scan_blocks_h;
    uint32_t* hist_off = hist + hist_cnt;
    uint32_t* idx = hist_off + hist_cnt + 1;
    uint64_t* entries = reinterpret_cast<uint64_t*>(ws + ((head_words + 3) & ~(size_t)3));
    const uint32_t red_pts = nb / 4 + 1024;       // scratch of the weighted bucket sum (group sums of every level, partials)
    const size_t max_tasks = (size_t)nb + std::max((size_t)(max_entries / TASK_CAP), (size_t)TASK_TARGET) + 1;
    const size_t npts29 = std::max((size_t)nb + red_pts + max_tasks, npts29_N);
    // three bucket buffers in rotation: the reduction of MSM it (a latency-bound chain on a side stream, about as long
    // as a whole MSM) must only be finished when MSM it + 3 starts to accumulate
    char* bkbuf[3];
    bkbuf[0] = (char*)ctx->get_scratch(SC_MSM_BUCKETS, sizeof(G1Xyzz29) * npts29);
    bkbuf[1] = count > 1 ? (char*)ctx->get_scratch(SC_MSM_BUCKETS2, sizeof(G1Xyzz29) * npts29) : bkbuf[0];
    bkbuf[2] = count > 2 ? (char*)ctx->get_scratch(SC_MSM_BUCKETS3, sizeof(G1Xyzz29) * npts29) : bkbuf[0];
    // one window sum per MSM, then one private copy of the window flags per MSM of the per-window path
    G1Xyzz* wsum_all = (G1Xyzz*)ctx->get_scratch(SC_MSM_RESULTS, sizeof(G1Xyzz) * count + 4 * MSM_WFLAGS * count);
    if (!bkbuf[0] || !bkbuf[1] || !bkbuf[2] || !wsum_all) return ZK_ERR_OOM;
    if (!ctx->stream2) ZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    if (!ctx->ev_p1[0])
        for (int i = 0; i < 3; ++i) {
            ZK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_p1[i], hipEventDisableTiming));
            ZK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_p2[i], hipEventDisableTiming));
        }
    // The reductions of consecutive MSMs rotate over three side streams: each is a chain of short
    // launches (latency, not throughput), and with witness columns that fill few windows it can take
    // longer than the sort + accumulation of the next column.
    if (!ctx->stream2b) ZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2b, hipStreamNonBlocking));
    if (!ctx->stream2c) ZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2c, hipStreamNonBlocking));
    if (!ctx->ev_pipe) ZK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_pipe, hipEventDisableTiming));
    // enqueues the tails kernel on the context's stream (all columns are on the device by then) and returns where its results land
    G1Xyzz* tails_dev = nullptr;
    auto enqueue_tails = [&]() -> int {
        if (!tail_rows) return ZK_OK;
        const size_t off_flags = (sizeof(void*) * count + 255) & ~(size_t)255, off_out = (off_flags + count + 255) & ~(size_t)255;
        char* tb = (char*)ctx->get_scratch(SC_MSM_TAILS, off_out + sizeof(G1Xyzz) * count);
        if (!tb) return ZK_ERR_OOM;
        ZK_HIP(ctx, hipMemcpyAsync(tb, d_scalar_ptrs, sizeof(void*) * count, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(ctx, hipMemcpyAsync(tb + off_flags, narrow, count, hipMemcpyHostToDevice, ctx->stream));
        tails_dev = (G1Xyzz*)(tb + off_out);
        hipLaunchKernelGGL(k_msm_tails, dim3((unsigned)count), dim3(1024), 0, ctx->stream, (const Fr* const*)tb, (const uint8_t*)(tb + off_flags), n_narrow, tail_rows,
                           d_table, (uint64_t)tab_stride, pl.c, pl.W, pl.top_shift, tails_dev);
        ZK_CHECK_LAUNCH(ctx);
        return ZK_OK;
    };
    if (want_graph) {
        // ---- graph mode: GP linear pipelines, column it on pipeline it % GP (stream, sort workspace and bucket buffer of its own;
        // the reduction runs on the same stream: one launch, short chain), one hipGraphLaunch per column.
        hipStream_t P[GP] = {ctx->stream, ctx->stream2, ctx->stream2b, ctx->stream2c};
        char* bk[GP] = {bkbuf[0], bkbuf[1], bkbuf[2], (char*)ctx->get_scratch(SC_MSM_BUCKETS4, sizeof(G1Xyzz29) * npts29)};
        char* desc = (char*)ctx->get_scratch(SC_MSM_DESC, sizeof(MsmCol) * count + 256);
        if (!bk[3] || !desc) return ZK_ERR_OOM;
        uint32_t* ctr_dev = (uint32_t*)desc;                                  // GP counters, then the table
        MsmCol* cols_dev = (MsmCol*)(desc + 256);
        {
            std::vector<MsmCol> hcols(count);
            for (size_t i = 0; i < count; ++i) hcols[i] = MsmCol{d_scalar_ptrs[i], wsum_all + i};
            const uint32_t hctr[GP] = {0, 1, 2, 3};
            ZK_HIP(ctx, hipMemcpyAsync(cols_dev, hcols.data(), sizeof(MsmCol) * count, hipMemcpyHostToDevice, P[0]));
            ZK_HIP(ctx, hipMemcpyAsync(ctr_dev, hctr, sizeof(hctr), hipMemcpyHostToDevice, P[0]));
            ZK_HIP(ctx, hipStreamSynchronize(P[0]));                          // the host copies live on this stack frame
        }
        ZK_HIP(ctx, hipEventRecord(ctx->ev_pipe, P[0]));
        for (int p = 1; p < GP; ++p) ZK_HIP(ctx, hipStreamWaitEvent(P[p], ctx->ev_pipe, 0));
        struct StreamRestoreG { zk_ctx* c; hipStream_t s; ~StreamRestoreG() { c->stream = s; } } restore_g{ctx, ctx->stream};
        // the launch sequence of one column of `kind` on pipeline p, enqueued on ctx->stream (being captured)
        auto enqueue = [&](int kind, int p) -> int {
            hipStream_t st = ctx->stream;
            uint32_t* wsb = ws + (size_t)p * words;
            G1Xyzz29* buckets = (G1Xyzz29*)bk[p];
            const uint32_t* ctr = ctr_dev + p;
            if (kind == 0) {           // per-window path over the narrow table
                uint32_t* slice_countsN = wsb;
                uint32_t* slice_offN = slice_countsN + (size_t)nbN * MSM_SLICES;
                uint32_t* countsN = slice_offN + (size_t)nbN * MSM_SLICES + 4;
                uint32_t* size_histN = countsN + nbN;
                uint32_t* nmultiN = size_histN + SIZE_BINS;
                uint32_t* wflag = nmultiN + 4;
                uint32_t* offsetsN = wflag + MSM_WFLAGS + TASK_DONE_MAX;
                uint32_t* orderN = offsetsN + nbN + 1;
                uint32_t* ntasksN = orderN + nbN;
                uint32_t* toffN = ntasksN + nbN;
                uint32_t* block_totN = toffN + nbN + 1;
                uint32_t* block_tot2N = block_totN + scan_blocks_sN;
                uint32_t* idxN = block_tot2N + scan_blocks_N;
                uint16_t* dig = reinterpret_cast<uint16_t*>(wsb + ((head_words_N + 3) & ~(size_t)3));
                G1Xyzz29* partialN = buckets + nbN;
                G1Xyzz29* task_partialN = partialN + (size_t)red_blocks_N * NG;
                G1Xyzz29* folded = task_partialN + max_tasks_N;
                const dim3 sweep_grid(8u * ((pn.W + 7) / 8) * (pn.B >> range_bits_N) * MSM_SLICES);
                ZK_HIP(ctx, hipMemsetAsync(size_histN, 0, (size_t)(SIZE_BINS + 4 + MSM_WFLAGS + TASK_DONE_MAX) * 4, st));
                launch_digits(pn.c, dim3((unsigned)((n_pad / 2 + 255) / 256)), st, (const Fr*)nullptr, n_narrow, n_pad, dig, wflag, cols_dev, ctr);
                hipLaunchKernelGGL((k_msm_lds_sweep<false>), sweep_grid, dim3(1024), 0, st, (const uint16_t*)dig, n_pad, range_bits_N, pn.B, slice_countsN, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)wflag, (uint32_t)pn.W);
                hipLaunchKernelGGL(k_scan_u32_a, dim3(scan_blocks_sN), dim3(SCAN_T), 0, st, (const uint32_t*)slice_countsN, nbN * MSM_SLICES, slice_offN, block_totN);
                hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, st, block_totN, scan_blocks_sN, slice_offN, nbN * MSM_SLICES, offsetsN + nbN);
                hipLaunchKernelGGL(k_scan_u32_c, dim3(scan_blocks_sN), dim3(SCAN_T), 0, st, (const uint32_t*)slice_countsN, nbN, slice_offN, (const uint32_t*)block_totN, offsetsN, countsN, size_histN);
                hipLaunchKernelGGL(k_size_bins_scan, dim3(1), dim3(64), 0, st, size_histN, (const uint32_t*)(offsetsN + nbN));
                hipLaunchKernelGGL(k_order_buckets, dim3(scan_blocks_N), dim3(SCAN_T), 0, st, (const uint32_t*)countsN, nbN, size_histN, orderN, ntasksN);
                hipLaunchKernelGGL(k_scan_u32_a, dim3(scan_blocks_N), dim3(SCAN_T), 0, st, (const uint32_t*)ntasksN, nbN, toffN, block_tot2N);
                hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, st, block_tot2N, scan_blocks_N, toffN, nbN, (uint32_t*)nullptr);
                hipLaunchKernelGGL(k_task_offsets, dim3(scan_blocks_N), dim3(SCAN_T), 0, st, nbN, toffN, (const uint32_t*)block_tot2N);
                hipLaunchKernelGGL((k_msm_lds_sweep<true>), sweep_grid, dim3(1024), 0, st, (const uint16_t*)dig, n_pad, range_bits_N, pn.B, (uint32_t*)nullptr, (const uint32_t*)slice_offN, idxN, (const uint32_t*)wflag, (uint32_t)pn.W);
                hipLaunchKernelGGL(k_msm_buckets, dim3((unsigned)((max_tasks_N + 255) / 256)), dim3(256), 0, st, d_table_n, (const uint32_t*)offsetsN, (const uint32_t*)idxN,
                                   (const uint32_t*)orderN, (const uint32_t*)toffN, (const uint32_t*)nmultiN, nbN, buckets, task_partialN, pn.c - 1, (uint64_t)tab_stride, (const uint32_t*)wflag);
                hipLaunchKernelGGL(k_msm_combine_wave, dim3(combine_grid((max_tasks_N + 255) / 256)), dim3(256), 0, st, (const uint32_t*)nmultiN, (const uint32_t*)toffN, task_partialN);
                hipLaunchKernelGGL(k_msm_combine_small, dim3(combine_grid((nbN + 255) / 256)), dim3(256), 0, st, (const uint32_t*)nmultiN, (const uint32_t*)orderN,
                                   (const uint32_t*)ntasksN, (const uint32_t*)toffN, (const G1Xyzz29*)task_partialN, buckets);
                hipLaunchKernelGGL(k_msm_combine, dim3(256), dim3(256), 0, st, (const uint32_t*)nmultiN, (const uint32_t*)orderN,
                                   (const uint32_t*)ntasksN, (const uint32_t*)toffN, (const G1Xyzz29*)task_partialN, buckets);
                // the window flags are still this column's when the fold reads them: the next column of this pipeline comes behind it on the same stream
                hipLaunchKernelGGL(k_msm_fold_windows, dim3((pn.B + 63) / 64), dim3(256), 0, st, (const G1Xyzz29*)buckets, pn.B, pn.W, folded, (const uint32_t*)wflag);
                hipLaunchKernelGGL((k_msm_reduce<RED_G_WIDE>), dim3(red_blocks_N, 1), dim3(RED_THREADS), 0, st, (const G1Xyzz29*)folded, pn.B, partialN);
                hipLaunchKernelGGL(k_msm_window_sum, dim3(1), dim3(RED_THREADS), 0, st, (const G1Xyzz29*)partialN, red_blocks_N, (G1Xyzz*)nullptr, cols_dev, ctr_dev + p, (uint32_t)GP);
                ZK_CHECK_LAUNCH(ctx);
                return ZK_OK;
            }
            // merged-window path (kind 1: one-launch partition sort, kind 2: sliced sort)
            uint32_t* slice_counts = wsb;
            uint32_t* slice_off = slice_counts + (size_t)nb * MSM_SLICES;
            uint32_t* counts = slice_off + (size_t)nb * MSM_SLICES + 4;
            uint32_t* size_hist = counts + nb;
            uint32_t* nmulti = size_hist + SIZE_BINS;
            uint32_t* offsets = nmulti + 4 + MSM_WFLAGS + TASK_DONE_MAX;
            uint32_t* order = offsets + nb + 1;
            uint32_t* ntasks = order + nb;
            uint32_t* toff = ntasks + nb;
            uint32_t* block_tot = toff + nb + 1;
            uint32_t* block_tot2 = block_tot + scan_blocks_s;
            uint32_t* block_tot3 = block_tot2 + scan_blocks;
            uint32_t* hist = block_tot3 + scan_blocks_h;
            uint32_t* hist_off = hist + hist_cnt;
            uint32_t* idx = hist_off + hist_cnt + 1;
            uint64_t* entries = reinterpret_cast<uint64_t*>(wsb + ((head_words + 3) & ~(size_t)3));
            G1Xyzz29* partial = buckets + nb;
            G1Xyzz29* task_partial = partial + red_pts;
            const bool bs_it = kind == 1;
            ZK_HIP(ctx, hipMemsetAsync(size_hist, 0, (size_t)(SIZE_BINS + 4 + MSM_WFLAGS + TASK_DONE_MAX) * 4, st));
            launch_partition<false>(pl.c, dim3(nwg), st, (const Fr*)nullptr, (uint64_t)n, range_bits, hist, (const uint32_t*)nullptr, (uint64_t*)nullptr, (uint64_t)tab_stride, pl.top_shift, cols_dev, ctr);
            hipLaunchKernelGGL(k_scan_u32_a, dim3(scan_blocks_h), dim3(SCAN_T), 0, st, (const uint32_t*)hist, hist_cnt, hist_off, block_tot3);
            hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, st, block_tot3, scan_blocks_h, hist_off, hist_cnt, (uint32_t*)nullptr);
            hipLaunchKernelGGL(k_task_offsets, dim3(scan_blocks_h), dim3(SCAN_T), 0, st, hist_cnt, hist_off, (const uint32_t*)block_tot3);
            if (staged_scatter) PK_TRY_MSM(launch_scatter_staged(ctx, pl.c, pl.W, dim3(nwg), (const Fr*)nullptr, (uint64_t)n, range_bits, (const uint32_t*)hist_off, entries, (uint64_t)tab_stride, pl.top_shift, cols_dev, ctr));
            else launch_partition<true>(pl.c, dim3(nwg), st, (const Fr*)nullptr, (uint64_t)n, range_bits, (uint32_t*)nullptr, (const uint32_t*)hist_off, entries, (uint64_t)tab_stride, pl.top_shift, cols_dev, ctr);
            if (bs_it) {
                hipLaunchKernelGGL(k_msm_m_binsort, dim3(nbins), dim3(1024), 0, st, (const uint64_t*)entries, (const uint32_t*)hist_off, nwg, range_bits, nb, offsets, counts, size_hist, idx);
            } else {
                hipLaunchKernelGGL((k_msm_m_bin<false>), dim3(nbins * MSM_SLICES), dim3(1024), 0, st, (const uint64_t*)entries, (const uint32_t*)hist_off, nwg, range_bits, slice_counts, (const uint32_t*)nullptr, (uint32_t*)nullptr);
                hipLaunchKernelGGL(k_scan_u32_a, dim3(scan_blocks_s), dim3(SCAN_T), 0, st, (const uint32_t*)slice_counts, nb * MSM_SLICES, slice_off, block_tot);
                hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, st, block_tot, scan_blocks_s, slice_off, nb * MSM_SLICES, offsets + nb);
                hipLaunchKernelGGL(k_scan_u32_c, dim3(scan_blocks_s), dim3(SCAN_T), 0, st, (const uint32_t*)slice_counts, nb, slice_off, (const uint32_t*)block_tot, offsets, counts, size_hist);
            }
            hipLaunchKernelGGL(k_size_bins_scan, dim3(1), dim3(64), 0, st, size_hist, (const uint32_t*)(offsets + nb));
            hipLaunchKernelGGL(k_order_buckets, dim3(scan_blocks), dim3(SCAN_T), 0, st, (const uint32_t*)counts, nb, size_hist, order, ntasks);
            hipLaunchKernelGGL(k_scan_u32_a, dim3(scan_blocks), dim3(SCAN_T), 0, st, (const uint32_t*)ntasks, nb, toff, block_tot2);
            hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, st, block_tot2, scan_blocks, toff, nb, (uint32_t*)nullptr);
            hipLaunchKernelGGL(k_task_offsets, dim3(scan_blocks), dim3(SCAN_T), 0, st, nb, toff, (const uint32_t*)block_tot2);
            if (!bs_it) hipLaunchKernelGGL((k_msm_m_bin<true>), dim3(nbins * MSM_SLICES), dim3(1024), 0, st, (const uint64_t*)entries, (const uint32_t*)hist_off, nwg, range_bits, (uint32_t*)nullptr, (const uint32_t*)slice_off, idx);
            hipLaunchKernelGGL(k_msm_buckets, dim3((unsigned)((max_tasks + 255) / 256)), dim3(256), 0, st, d_table, (const uint32_t*)offsets, (const uint32_t*)idx,
                               (const uint32_t*)order, (const uint32_t*)toff, (const uint32_t*)nmulti, nb, buckets, task_partial, 0, (uint64_t)0, (const uint32_t*)nullptr);
            hipLaunchKernelGGL(k_msm_combine_wave, dim3(combine_grid((max_tasks + 255) / 256)), dim3(256), 0, st, (const uint32_t*)nmulti, (const uint32_t*)toff, task_partial);
            hipLaunchKernelGGL(k_msm_combine_small, dim3(combine_grid((nb + 255) / 256)), dim3(256), 0, st, (const uint32_t*)nmulti, (const uint32_t*)order,
                               (const uint32_t*)ntasks, (const uint32_t*)toff, (const G1Xyzz29*)task_partial, buckets);
            hipLaunchKernelGGL(k_msm_combine, dim3(256), dim3(256), 0, st, (const uint32_t*)nmulti, (const uint32_t*)order,
                               (const uint32_t*)ntasks, (const uint32_t*)toff, (const G1Xyzz29*)task_partial, buckets);
            const uint32_t rb = ((nb + RED_G_WIDE - 1) / RED_G_WIDE + RED_THREADS - 1) / RED_THREADS;
            hipLaunchKernelGGL((k_msm_reduce<RED_G_WIDE>), dim3(rb, 1), dim3(RED_THREADS), 0, st, (const G1Xyzz29*)buckets, nb, partial);
            hipLaunchKernelGGL(k_msm_window_sum, dim3(1), dim3(RED_THREADS), 0, st, (const G1Xyzz29*)partial, rb, (G1Xyzz*)nullptr, cols_dev, ctr_dev + p, (uint32_t)GP);
            ZK_CHECK_LAUNCH(ctx);
            return ZK_OK;
        };
        if (stage) { int rc = stage(stage_user, 0); if (rc) return rc; }
        bool fallback = false;
        for (size_t it = 0; it < count && !fallback; ++it) {
            const int p = (int)(it % GP);
            const int kind = (any_narrow && narrow[it] == 1) ? 0 : ((binsort && !(narrow && narrow[it] == 2)) ? 1 : 2);
            ctx->stream = P[p];
            // a graph is tied to every address baked into its kernel arguments
            uint64_t key = 1469598103934665603ull;
            for (uint64_t v : {(uint64_t)kind, (uint64_t)p, (uint64_t)n, (uint64_t)pl.c, (uint64_t)pn.c, (uint64_t)(uintptr_t)d_table, (uint64_t)(uintptr_t)d_table_n, (uint64_t)tab_stride,
                               (uint64_t)(uintptr_t)ws, (uint64_t)words, (uint64_t)(uintptr_t)bk[p], (uint64_t)(uintptr_t)desc, (uint64_t)staged_scatter, (uint64_t)range_bits, n_narrow})
                key = (key ^ v) * 1099511628211ull;
            hipGraphExec_t exec = nullptr;
            const std::vector<uint64_t> key_tuple = {(uint64_t)kind, (uint64_t)p, (uint64_t)n, (uint64_t)pl.c, (uint64_t)pn.c, (uint64_t)(uintptr_t)d_table, (uint64_t)(uintptr_t)d_table_n, (uint64_t)tab_stride,
                                                     (uint64_t)(uintptr_t)ws, (uint64_t)words, (uint64_t)(uintptr_t)bk[p], (uint64_t)(uintptr_t)desc, (uint64_t)staged_scatter, (uint64_t)range_bits, n_narrow};
            auto found = ctx->msm_graphs.find(key);
            if (found != ctx->msm_graphs.end() && ctx->msm_graph_keys[key] != key_tuple) {       // a 64-bit hash collision: the stale graph goes
                (void)hipGraphExecDestroy((hipGraphExec_t)found->second);
                ctx->msm_graphs.erase(found);
                found = ctx->msm_graphs.end();
            }
            if (found != ctx->msm_graphs.end()) exec = (hipGraphExec_t)found->second;
            else {
                // graphs bake scratch addresses in; when the scratch arenas were reallocated since, the old entries can never be
                // hit again: the cache is emptied when it grows past what one key / size mix needs
                if (ctx->msm_graphs.size() >= 48) {
                    ZK_HIP(ctx, hipStreamSynchronize(P[0]));
                    for (int q = 1; q < GP; ++q) ZK_HIP(ctx, hipStreamSynchronize(P[q]));
                    for (auto& kv : ctx->msm_graphs) (void)hipGraphExecDestroy((hipGraphExec_t)kv.second);
                    ctx->msm_graphs.clear();
                    ctx->msm_graph_keys.clear();
                }
                hipGraph_t graph = nullptr;
                bool ok = hipStreamBeginCapture(P[p], hipStreamCaptureModeRelaxed) == hipSuccess;
                int rc_e = ok ? enqueue(kind, p) : ZK_ERR_HIP;
                if (ok) ok = hipStreamEndCapture(P[p], &graph) == hipSuccess && graph != nullptr;
                ok = ok && rc_e == ZK_OK && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
                if (graph) (void)hipGraphDestroy(graph);
                if (!ok) { (void)hipGetLastError(); ctx->msm_graph_broken = true; fallback = true; break; }
                ctx->msm_graphs[key] = (void*)exec;
                ctx->msm_graph_keys[key] = key_tuple;
            }
            if (hipGraphLaunch(exec, P[p]) != hipSuccess) { (void)hipGetLastError(); ctx->msm_graph_broken = true; fallback = true; break; }
            if (stage && it + 1 < count) { ctx->stream = P[(it + 1) % GP]; int rc = stage(stage_user, it + 1); if (rc) return rc; }
        }
        ctx->stream = P[0];
        for (int p = 1; p < GP; ++p) {
            ZK_HIP(ctx, hipEventRecord(ctx->ev_p2[p - 1], P[p]));
            ZK_HIP(ctx, hipStreamWaitEvent(P[0], ctx->ev_p2[p - 1], 0));
        }
        if (!fallback) {
            PK_TRY_MSM(enqueue_tails());
            std::vector<G1Xyzz> hw(count), ht(tail_rows ? count : 0);
            if (tail_rows) ZK_HIP(ctx, hipMemcpyAsync(ht.data(), tails_dev, sizeof(G1Xyzz) * count, hipMemcpyDeviceToHost, ctx->stream));
            ZK_HIP(ctx, hipMemcpyAsync(hw.data(), wsum_all, sizeof(G1Xyzz) * count, hipMemcpyDeviceToHost, ctx->stream));
            const auto t_enq = std::chrono::steady_clock::now();
            ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (getenv("ZK_MSM_TRACE")) {
                const auto t_done = std::chrono::steady_clock::now();
                fprintf(stderr, "[zk msm] batch of %zu x 2^%.0f (graph replay): host enqueue %.3f ms, device drained %.3f ms later\n", count, log2((double)n),
                        std::chrono::duration<double, std::milli>(t_enq - t_batch0).count(), std::chrono::duration<double, std::milli>(t_done - t_enq).count());
            }
            for (size_t it = 0; it < count; ++it) {
                if (tail_rows && narrow[it] == 1) host::msm_tail2(hw.data() + it, ht.data() + it, h_out + it);
                else host::msm_tail(hw.data() + it, 1, pl.c, h_out + it);
            }
            return ZK_OK;
        }
        // capture or replay failed: drain what was launched and take the plain path below for the whole batch (a staging
        // callback is simply asked again: uploading a column twice and redoing its transform are idempotent)
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    hipStream_t mains[2] = {ctx->stream, npipe == 2 ? ctx->stream2c : ctx->stream};
    struct StreamRestore { zk_ctx* c; hipStream_t s; ~StreamRestore() { c->stream = s; } } restore{ctx, ctx->stream};     // error returns included
    if (npipe == 2) {           // the second pipeline starts behind everything already enqueued on the context's stream
        ZK_HIP(ctx, hipEventRecord(ctx->ev_pipe, mains[0]));
        ZK_HIP(ctx, hipStreamWaitEvent(mains[1], ctx->ev_pipe, 0));
    }
    // Sort-ahead (one pipeline, >= 2 columns): the sort of MSM it + 1 (recoding, partition, counting sort, task split: a third
    // of an MSM's time on the main stream, bound by memory and launch latency) runs on the auxiliary stream UNDER the bucket
    // accumulation of MSM it (bound by integer issue), on the second copy of the sort workspace.  OFF by default
    // (ZK_MSM_SORT_AHEAD=1 enables): measured in round 3 (profiles/r03_sort_ahead.md) it LOSES 11-15 % -- beside a kernel with
    // thousands of workgroups pending, the sort's 1024-thread / 104 KiB-LDS workgroups rarely find a CU with room, its dozen
    // dependent launches stretch from 0.29 to 0.8-1.3 ms, and the accumulation slows by the slots they do get; a
    // high-priority stream changes nothing.  Kept as a knob because the refactoring it needed (sort / accumulate as separate
    // steps over workspace slots) is what a future partition-level pipeline would start from.
    const bool sort_ahead = npipe == 1 && ws_copies == 2;
    hipStream_t sort_st = nullptr;
    if (sort_ahead) {
        if (!ctx->ensure_aux()) return ctx->fail(ZK_ERR_HIP, "could not create the auxiliary stream");
        sort_st = ctx->stream_aux;
        for (int i = 0; i < 2; ++i) if (!ctx->ev_sorted[i]) ZK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_sorted[i], hipEventDisableTiming));
        ZK_HIP(ctx, hipEventRecord(ctx->ev_pipe, mains[0]));          // the sorts start behind everything already enqueued on the context's stream
        ZK_HIP(ctx, hipStreamWaitEvent(sort_st, ctx->ev_pipe, 0));
    }
    // the sort workspace of a slot, merged-window layout (see above) and per-window layout
    struct WsM { uint32_t *slice_counts, *slice_off, *counts, *size_hist, *nmulti, *offsets, *order, *ntasks, *toff, *block_tot, *block_tot2, *block_tot3, *hist, *hist_off, *idx; uint64_t* entries; };
    struct WsN { uint32_t *slice_counts, *slice_off, *counts, *size_hist, *nmulti, *wflag, *offsets, *order, *ntasks, *toff, *block_tot, *block_tot2, *idx; uint16_t* dig; };
    auto ws_merged = [&](int slot) {
        WsM w;
        uint32_t* wsb = ws + (size_t)slot * words;
        w.slice_counts = wsb;
        w.slice_off = w.slice_counts + (size_t)nb * MSM_SLICES;
        w.counts = w.slice_off + (size_t)nb * MSM_SLICES + 4;
        w.size_hist = w.counts + nb;
        w.nmulti = w.size_hist + SIZE_BINS;
        w.offsets = w.nmulti + 4 + MSM_WFLAGS + TASK_DONE_MAX;
        w.order = w.offsets + nb + 1;
        w.ntasks = w.order + nb;
        w.toff = w.ntasks + nb;
        w.block_tot = w.toff + nb + 1;
        w.block_tot2 = w.block_tot + scan_blocks_s;
        w.block_tot3 = w.block_tot2 + scan_blocks;
        w.hist = w.block_tot3 + scan_blocks_h;
        w.hist_off = w.hist + hist_cnt;
        w.idx = w.hist_off + hist_cnt + 1;
        w.entries = reinterpret_cast<uint64_t*>(wsb + ((head_words + 3) & ~(size_t)3));
        return w;
    };
    auto ws_narrow = [&](int slot) {
        WsN w;
        uint32_t* wsb = ws + (size_t)slot * words;
        w.slice_counts = wsb;
        w.slice_off = w.slice_counts + (size_t)nbN * MSM_SLICES;
        w.counts = w.slice_off + (size_t)nbN * MSM_SLICES + 4;
        w.size_hist = w.counts + nbN;
        w.nmulti = w.size_hist + SIZE_BINS;
        w.wflag = w.nmulti + 4;
        w.offsets = w.wflag + MSM_WFLAGS + TASK_DONE_MAX;
        w.order = w.offsets + nbN + 1;
        w.ntasks = w.order + nbN;
        w.toff = w.ntasks + nbN;
        w.block_tot = w.toff + nbN + 1;
        w.block_tot2 = w.block_tot + scan_blocks_sN;
        w.idx = w.block_tot2 + scan_blocks_N;
        w.dig = reinterpret_cast<uint16_t*>(wsb + ((head_words_N + 3) & ~(size_t)3));
        return w;
    };
    struct WsG { uint32_t *counts, *size_hist, *nmulti, *offsets, *order, *ntasks, *toff, *block_tot2, *block_tot3, *hist, *hist_off, *idx; uint64_t* entries; };
    auto ws_gm = [&](int slot) {
        WsG w;
        uint32_t* wsb = ws + (size_t)slot * words;
        w.counts = wsb;
        w.size_hist = w.counts + nbG;
        w.nmulti = w.size_hist + SIZE_BINS;
        w.offsets = w.nmulti + 4 + MSM_WFLAGS + TASK_DONE_MAX;
        w.order = w.offsets + nbG + 1;
        w.ntasks = w.order + nbG;
        w.toff = w.ntasks + nbG;
        w.block_tot2 = w.toff + nbG + 1;
        w.block_tot3 = w.block_tot2 + scan_blocks_G;
        w.hist = w.block_tot3 + scan_blocks_hG;
        w.hist_off = w.hist + hist_cnt_G;
        w.idx = w.hist_off + hist_cnt_G + 1;
        w.entries = reinterpret_cast<uint64_t*>(wsb + ((head_words_G + 3) & ~(size_t)3));
        return w;
    };
    const dim3 sweep_grid(8u * ((pn.W + 7) / 8) * (pn.B >> range_bits_N) * MSM_SLICES);
    auto is_narrow = [&](size_t it) { return any_narrow && narrow[it] == 1; };
    // ---- the sort of MSM `it` into workspace `slot`, enqueued on `st`
    // the group of columns column `it` starts: up to NG consecutive small-valued columns, or the column alone
    auto group_of = [&](size_t it) -> size_t {
        if (!is_narrow(it)) return 1;
        size_t g = 1;
        while (g < NG && it + g < count && is_narrow(it + g)) ++g;
        return g;
    };
    auto enqueue_sort = [&](size_t it, int slot, hipStream_t st) -> int {
        const Fr* d_scalars = d_scalar_ptrs[it];
        ZkProfScope ps(ctx, "msm_sort", st);
        if (is_narrow(it) && gm) {
            // GM sort of the group (see above): per column a histogram and a scatter pass over its scalars into the group's shared
            // partition space, then ONE counting sort, size ordering and task split over the cnt * B buckets of the group
            const uint32_t cnt = (uint32_t)group_of(it), nbc = cnt * SG * pn.B, nbins_c = cnt * bpcG, hist_c = nbins_c * nwgG;
            const uint32_t sb = (nbc + SCAN_T * SCAN_ITEMS - 1) / (SCAN_T * SCAN_ITEMS), sb_h = (hist_c + SCAN_T * SCAN_ITEMS - 1) / (SCAN_T * SCAN_ITEMS);
            const WsG w = ws_gm(slot);
            ZK_HIP(ctx, hipMemsetAsync(w.size_hist, 0, (size_t)(SIZE_BINS + 4 + MSM_WFLAGS + TASK_DONE_MAX) * 4, st));
            GmCols gcols{};
            for (uint32_t j = 0; j < cnt; ++j) gcols.p[j] = d_scalar_ptrs[it + j];
            launch_gm_partition<false>(pn.c, dim3(nwgG, cnt), st, gcols, n_narrow, range_bits_G, w.hist, (const uint32_t*)nullptr, (uint64_t*)nullptr, (uint64_t)tab_stride, bpcG, SG - 1);
            ZK_CHECK_LAUNCH(ctx);
            hipLaunchKernelGGL(k_scan_u32_a, dim3(sb_h), dim3(SCAN_T), 0, st, (const uint32_t*)w.hist, hist_c, w.hist_off, w.block_tot3);
            hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, st, w.block_tot3, sb_h, w.hist_off, hist_c, (uint32_t*)nullptr);
            hipLaunchKernelGGL(k_task_offsets, dim3(sb_h), dim3(SCAN_T), 0, st, hist_c, w.hist_off, (const uint32_t*)w.block_tot3);
            launch_gm_partition<true>(pn.c, dim3(nwgG, cnt), st, gcols, n_narrow, range_bits_G, (uint32_t*)nullptr, (const uint32_t*)w.hist_off, w.entries, (uint64_t)tab_stride, bpcG, SG - 1);
            ZK_CHECK_LAUNCH(ctx);
            hipLaunchKernelGGL(k_msm_m_binsort, dim3(nbins_c), dim3(1024), 0, st, (const uint64_t*)w.entries, (const uint32_t*)w.hist_off, nwgG, range_bits_G, nbc, w.offsets, w.counts, w.size_hist, w.idx);
            ZK_CHECK_LAUNCH(ctx);
            hipLaunchKernelGGL(k_size_bins_scan, dim3(1), dim3(64), 0, st, w.size_hist, (const uint32_t*)(w.offsets + nbc));
            hipLaunchKernelGGL(k_order_buckets, dim3(sb), dim3(SCAN_T), 0, st, (const uint32_t*)w.counts, nbc, w.size_hist, w.order, w.ntasks);
            hipLaunchKernelGGL(k_scan_u32_a, dim3(sb), dim3(SCAN_T), 0, st, (const uint32_t*)w.ntasks, nbc, w.toff, w.block_tot2);
            hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, st, w.block_tot2, sb, w.toff, nbc, (uint32_t*)nullptr);
            hipLaunchKernelGGL(k_task_offsets, dim3(sb), dim3(SCAN_T), 0, st, nbc, w.toff, (const uint32_t*)w.block_tot2);
            ZK_CHECK_LAUNCH(ctx);
            return ZK_OK;
        }
        if (is_narrow(it)) {
            // per-window path over the narrow table (see msm_batch_tab): digits, LDS-privatised sort with empty windows skipped;
            // the columns of a group are windows [j W, (j + 1) W) of one sort
            const uint32_t cnt = (uint32_t)group_of(it), wins = cnt * (uint32_t)pn.W, nbc = wins * pn.B;      // this group's windows and buckets
            const uint32_t sb_s = (nbc + SCAN_T - 1) / SCAN_T, sb = (nbc + SCAN_T * SCAN_ITEMS - 1) / (SCAN_T * SCAN_ITEMS);
            const dim3 grid_sw(8u * ((wins + 7) / 8) * (pn.B >> range_bits_N) * MSM_SLICES);
            const WsN w = ws_narrow(slot);
            uint32_t* wflag_it = reinterpret_cast<uint32_t*>(wsum_all + count) + it * MSM_WFLAGS;
            ZK_HIP(ctx, hipMemsetAsync(w.size_hist, 0, (size_t)(SIZE_BINS + 4 + MSM_WFLAGS + TASK_DONE_MAX) * 4, st));
            for (uint32_t j = 0; j < cnt; ++j)
                launch_digits(pn.c, dim3((unsigned)((n_pad / 2 + 255) / 256)), st, d_scalar_ptrs[it + j], n_narrow, n_pad, w.dig + (size_t)j * pn.W * n_pad, w.wflag + j * pn.W);
            hipLaunchKernelGGL((k_msm_lds_sweep<false>), grid_sw, dim3(1024), 0, st, (const uint16_t*)w.dig, n_pad, range_bits_N, pn.B, w.slice_counts, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)w.wflag, wins);
            ZK_CHECK_LAUNCH(ctx);
            hipLaunchKernelGGL(k_scan_u32_a, dim3(sb_s), dim3(SCAN_T), 0, st, (const uint32_t*)w.slice_counts, nbc * MSM_SLICES, w.slice_off, w.block_tot);
            hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, st, w.block_tot, sb_s, w.slice_off, nbc * MSM_SLICES, w.offsets + nbc);
            hipLaunchKernelGGL(k_scan_u32_c, dim3(sb_s), dim3(SCAN_T), 0, st, (const uint32_t*)w.slice_counts, nbc, w.slice_off, (const uint32_t*)w.block_tot, w.offsets, w.counts, w.size_hist);
            hipLaunchKernelGGL(k_size_bins_scan, dim3(1), dim3(64), 0, st, w.size_hist, (const uint32_t*)(w.offsets + nbc));
            hipLaunchKernelGGL(k_order_buckets, dim3(sb), dim3(SCAN_T), 0, st, (const uint32_t*)w.counts, nbc, w.size_hist, w.order, w.ntasks);
            hipLaunchKernelGGL(k_scan_u32_a, dim3(sb), dim3(SCAN_T), 0, st, (const uint32_t*)w.ntasks, nbc, w.toff, w.block_tot2);
            hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, st, w.block_tot2, sb, w.toff, nbc, (uint32_t*)nullptr);
            hipLaunchKernelGGL(k_task_offsets, dim3(sb), dim3(SCAN_T), 0, st, nbc, w.toff, (const uint32_t*)w.block_tot2);
            ZK_CHECK_LAUNCH(ctx);
            hipLaunchKernelGGL((k_msm_lds_sweep<true>), grid_sw, dim3(1024), 0, st, (const uint16_t*)w.dig, n_pad, range_bits_N, pn.B, (uint32_t*)nullptr, (const uint32_t*)w.slice_off, w.idx, (const uint32_t*)w.wflag, wins);
            ZK_CHECK_LAUNCH(ctx);
            ZK_HIP(ctx, hipMemcpyAsync(wflag_it, w.wflag, MSM_WFLAGS * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
            return ZK_OK;
        }
        const WsM w = ws_merged(slot);
        ZK_HIP(ctx, hipMemsetAsync(w.size_hist, 0, (size_t)(SIZE_BINS + 4 + MSM_WFLAGS + TASK_DONE_MAX) * 4, st));
        // 1. partition by the high bucket bits while recoding: histogram, scan, scatter
        launch_partition<false>(pl.c, dim3(nwg), st, d_scalars, (uint64_t)n, range_bits, w.hist, (const uint32_t*)nullptr, (uint64_t*)nullptr, (uint64_t)tab_stride, pl.top_shift);
        ZK_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL(k_scan_u32_a, dim3(scan_blocks_h), dim3(SCAN_T), 0, st, (const uint32_t*)w.hist, hist_cnt, w.hist_off, w.block_tot3);
        hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, st, w.block_tot3, scan_blocks_h, w.hist_off, hist_cnt, (uint32_t*)nullptr);
        hipLaunchKernelGGL(k_task_offsets, dim3(scan_blocks_h), dim3(SCAN_T), 0, st, hist_cnt, w.hist_off, (const uint32_t*)w.block_tot3);
        if (staged_scatter) PK_TRY_MSM(launch_scatter_staged(ctx, pl.c, pl.W, dim3(nwg), d_scalars, (uint64_t)n, range_bits, (const uint32_t*)w.hist_off, w.entries, (uint64_t)tab_stride, pl.top_shift, nullptr, nullptr, st));
        else launch_partition<true>(pl.c, dim3(nwg), st, d_scalars, (uint64_t)n, range_bits, (uint32_t*)nullptr, (const uint32_t*)w.hist_off, w.entries, (uint64_t)tab_stride, pl.top_shift);
        ZK_CHECK_LAUNCH(ctx);
        // 2. counting sort inside every partition, one launch (bucket offsets, counts, size histogram, sorted table indices);
        //    ZK_MSM_BINSORT=0 keeps the sliced count / scan / scatter sequence (measurement knob)
        const bool bs_it = binsort && !(narrow && narrow[it] == 2);       // hint 2: long runs of equal scalars (running products) put whole runs into
                                                                          // one partition; four slice-workgroups per partition stream them faster than one
        if (bs_it) {
            hipLaunchKernelGGL(k_msm_m_binsort, dim3(nbins), dim3(1024), 0, st, (const uint64_t*)w.entries, (const uint32_t*)w.hist_off, nwg, range_bits, nb, w.offsets, w.counts, w.size_hist, w.idx);
            ZK_CHECK_LAUNCH(ctx);
        } else {
            hipLaunchKernelGGL((k_msm_m_bin<false>), dim3(nbins * MSM_SLICES), dim3(1024), 0, st, (const uint64_t*)w.entries, (const uint32_t*)w.hist_off, nwg, range_bits, w.slice_counts, (const uint32_t*)nullptr, (uint32_t*)nullptr);
            ZK_CHECK_LAUNCH(ctx);
            hipLaunchKernelGGL(k_scan_u32_a, dim3(scan_blocks_s), dim3(SCAN_T), 0, st, (const uint32_t*)w.slice_counts, nb * MSM_SLICES, w.slice_off, w.block_tot);
            hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, st, w.block_tot, scan_blocks_s, w.slice_off, nb * MSM_SLICES, w.offsets + nb);
            hipLaunchKernelGGL(k_scan_u32_c, dim3(scan_blocks_s), dim3(SCAN_T), 0, st, (const uint32_t*)w.slice_counts, nb, w.slice_off, (const uint32_t*)w.block_tot, w.offsets, w.counts, w.size_hist);
        }
        hipLaunchKernelGGL(k_size_bins_scan, dim3(1), dim3(64), 0, st, w.size_hist, (const uint32_t*)(w.offsets + nb));
        hipLaunchKernelGGL(k_order_buckets, dim3(scan_blocks), dim3(SCAN_T), 0, st, (const uint32_t*)w.counts, nb, w.size_hist, w.order, w.ntasks);
        hipLaunchKernelGGL(k_scan_u32_a, dim3(scan_blocks), dim3(SCAN_T), 0, st, (const uint32_t*)w.ntasks, nb, w.toff, w.block_tot2);
        hipLaunchKernelGGL(k_scan_u32_b, dim3(1), dim3(SCAN_T), 0, st, w.block_tot2, scan_blocks, w.toff, nb, (uint32_t*)nullptr);
        hipLaunchKernelGGL(k_task_offsets, dim3(scan_blocks), dim3(SCAN_T), 0, st, nb, w.toff, (const uint32_t*)w.block_tot2);
        ZK_CHECK_LAUNCH(ctx);
        if (!bs_it) hipLaunchKernelGGL((k_msm_m_bin<true>), dim3(nbins * MSM_SLICES), dim3(1024), 0, st, (const uint64_t*)w.entries, (const uint32_t*)w.hist_off, nwg, range_bits, (uint32_t*)nullptr, (const uint32_t*)w.slice_off, w.idx);
        ZK_CHECK_LAUNCH(ctx);
        return ZK_OK;
    };
    if (stage) { int rc = stage(stage_user, 0); if (rc) return rc; }
    if (sort_ahead) {
        // whatever the staging callback made the context's stream wait for (the upload of column 0) the sort stream must see too
        ZK_HIP(ctx, hipEventRecord(ctx->ev_pipe, mains[0]));
        ZK_HIP(ctx, hipStreamWaitEvent(sort_st, ctx->ev_pipe, 0));
        PK_TRY_MSM(enqueue_sort(0, 0, sort_st));
        ZK_HIP(ctx, hipEventRecord(ctx->ev_sorted[0], sort_st));
    }
    size_t staged = stage ? 1 : count;            // columns [0, staged) have been handed to the staging callback
    size_t stepno = 0;                            // groups / columns enqueued so far: rotates bucket buffers, side streams, pipelines
    for (size_t it = 0; it < count; ++stepno) {
        const size_t grp = sort_ahead ? 1 : group_of(it);          // columns this step commits (a group of small-valued columns, or one column)
        const int par = (int)(stepno % 3), pipe = (int)(stepno % (size_t)npipe);
        const int slot = sort_ahead ? (int)(it & 1) : pipe;
        hipStream_t side = npipe == 2 ? ((stepno & 1) ? ctx->stream2b : ctx->stream2) : (par == 0 ? ctx->stream2 : par == 1 ? ctx->stream2b : ctx->stream2c);
        ctx->stream = mains[pipe];
        for (; staged < it + grp; ++staged) { int rc = stage(stage_user, staged); if (rc) return rc; }      // every column of the group is on its way before its sort is enqueued
        G1Xyzz29* buckets = (G1Xyzz29*)bkbuf[par];
        if (sort_ahead) ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_sorted[slot], 0));
        else PK_TRY_MSM(enqueue_sort(it, slot, ctx->stream));
        // reduce(step - 3) must be done with this bucket buffer -- but only the accumulation writes it: the sort of this MSM
        // runs while that reduction finishes
        if (stepno >= 3) ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_p2[par], 0));
        if (is_narrow(it) && gm) {
            // ---- GM: one bucket set per column, accumulated like a merged MSM (idx holds table indices); reduction and window sum per column
            const uint32_t cnt = (uint32_t)grp, nbc = cnt * SG * pn.B;
            const size_t tasks_c = (size_t)nbc + std::max(((size_t)n * cnt * pn.W) / TASK_CAP, (size_t)TASK_TARGET) + 1;
            const WsG w = ws_gm(slot);
            G1Xyzz29* partialN = buckets + nbN;
            G1Xyzz29* task_partialN = partialN + (size_t)red_blocks_N * NG * 4;
            {
                ZkProfScope ps(ctx, "msm_buckets_narrow");
                hipLaunchKernelGGL(k_msm_buckets, dim3((unsigned)((tasks_c + 255) / 256)), dim3(256), 0, ctx->stream, d_table_n, (const uint32_t*)w.offsets, (const uint32_t*)w.idx,
                                   (const uint32_t*)w.order, (const uint32_t*)w.toff, (const uint32_t*)w.nmulti, nbc, buckets, task_partialN, 0, (uint64_t)0, (const uint32_t*)nullptr, 0u, perm_G);
            }
            {
                ZkProfScope ps(ctx, "msm_combine");
                hipLaunchKernelGGL(k_msm_combine_wave, dim3(combine_grid((tasks_c + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)w.nmulti, (const uint32_t*)w.toff, task_partialN);
                hipLaunchKernelGGL(k_msm_combine_small, dim3(combine_grid((nbc + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)w.nmulti, (const uint32_t*)w.order,
                                   (const uint32_t*)w.ntasks, (const uint32_t*)w.toff, (const G1Xyzz29*)task_partialN, buckets, perm_G);
                hipLaunchKernelGGL(k_msm_combine, dim3(256), dim3(256), 0, ctx->stream, (const uint32_t*)w.nmulti, (const uint32_t*)w.order,
                                   (const uint32_t*)w.ntasks, (const uint32_t*)w.toff, (const G1Xyzz29*)task_partialN, buckets, perm_G);
                ZK_CHECK_LAUNCH(ctx);
            }
            ZK_HIP(ctx, hipEventRecord(ctx->ev_p1[par], ctx->stream));
            ZK_HIP(ctx, hipStreamWaitEvent(side, ctx->ev_p1[par], 0));
            {
                ZkProfScope ps(ctx, "msm_reduce_narrow", side);
                // the SG sets of a column are SG "windows" of equal weight: reduced separately, their partials summed with the column's
                hipLaunchKernelGGL((k_msm_reduce<RED_G_WIDE>), dim3(red_blocks_N, cnt * SG), dim3(RED_THREADS), 0, side, (const G1Xyzz29*)buckets, pn.B, partialN);
                ZK_CHECK_LAUNCH(ctx);
                hipLaunchKernelGGL(k_msm_window_sum, dim3(cnt), dim3(RED_THREADS), 0, side, (const G1Xyzz29*)partialN, red_blocks_N * SG, wsum_all + it);
                ZK_CHECK_LAUNCH(ctx);
            }
            ZK_HIP(ctx, hipEventRecord(ctx->ev_p2[par], side));
        } else if (is_narrow(it)) {
            // ---- per-window path: bucket accumulation, fold of the occupied windows of every column of the group, one window reduced per column
            const uint32_t cnt = (uint32_t)grp, wins = cnt * (uint32_t)pn.W, nbc = wins * pn.B;
            const size_t tasks_c = (size_t)nbc + std::max(((size_t)n * wins) / TASK_CAP, (size_t)TASK_TARGET) + 1;
            const WsN w = ws_narrow(slot);
            G1Xyzz29* partialN = buckets + nbN;
            G1Xyzz29* task_partialN = partialN + (size_t)red_blocks_N * NG;
            G1Xyzz29* folded = task_partialN + max_tasks_N;
            uint32_t* wflag_it = reinterpret_cast<uint32_t*>(wsum_all + count) + it * MSM_WFLAGS;
            {
                ZkProfScope ps(ctx, "msm_buckets_narrow");
                hipLaunchKernelGGL(k_msm_buckets, dim3((unsigned)((tasks_c + 255) / 256)), dim3(256), 0, ctx->stream, d_table_n, (const uint32_t*)w.offsets, (const uint32_t*)w.idx,
                                   (const uint32_t*)w.order, (const uint32_t*)w.toff, (const uint32_t*)w.nmulti, nbc, buckets, task_partialN, pn.c - 1, (uint64_t)tab_stride, (const uint32_t*)wflag_it,
                                   cnt > 1 ? (uint32_t)pn.W : 0u);
            }
            {
                ZkProfScope ps(ctx, "msm_combine");
                hipLaunchKernelGGL(k_msm_combine_wave, dim3(combine_grid((tasks_c + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)w.nmulti, (const uint32_t*)w.toff, task_partialN);
                hipLaunchKernelGGL(k_msm_combine_small, dim3(combine_grid((nbc + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)w.nmulti, (const uint32_t*)w.order,
                                   (const uint32_t*)w.ntasks, (const uint32_t*)w.toff, (const G1Xyzz29*)task_partialN, buckets);
                hipLaunchKernelGGL(k_msm_combine, dim3(256), dim3(256), 0, ctx->stream, (const uint32_t*)w.nmulti, (const uint32_t*)w.order,
                                   (const uint32_t*)w.ntasks, (const uint32_t*)w.toff, (const G1Xyzz29*)task_partialN, buckets);
                ZK_CHECK_LAUNCH(ctx);
            }
            ZK_HIP(ctx, hipEventRecord(ctx->ev_p1[par], ctx->stream));
            ZK_HIP(ctx, hipStreamWaitEvent(side, ctx->ev_p1[par], 0));
            {
                ZkProfScope ps(ctx, "msm_reduce_narrow", side);
                hipLaunchKernelGGL(k_msm_fold_windows, dim3((pn.B + 63) / 64, cnt), dim3(256), 0, side, (const G1Xyzz29*)buckets, pn.B, pn.W, folded, (const uint32_t*)wflag_it);
                hipLaunchKernelGGL((k_msm_reduce<RED_G_WIDE>), dim3(red_blocks_N, cnt), dim3(RED_THREADS), 0, side, (const G1Xyzz29*)folded, pn.B, partialN);
                ZK_CHECK_LAUNCH(ctx);
                hipLaunchKernelGGL(k_msm_window_sum, dim3(cnt), dim3(RED_THREADS), 0, side, (const G1Xyzz29*)partialN, red_blocks_N, wsum_all + it);
                ZK_CHECK_LAUNCH(ctx);
            }
            ZK_HIP(ctx, hipEventRecord(ctx->ev_p2[par], side));
        } else {
            const WsM w = ws_merged(slot);
            G1Xyzz29* partial = buckets + nb;
            G1Xyzz29* task_partial = partial + red_pts;
            {
                ZkProfScope ps(ctx, "msm_buckets");
                // idx already holds table indices: no window offset, no per-window skip (tab_stride = 0)
                hipLaunchKernelGGL(k_msm_buckets, dim3((unsigned)((max_tasks + 255) / 256)), dim3(256), 0, ctx->stream, d_table, (const uint32_t*)w.offsets, (const uint32_t*)w.idx,
                                   (const uint32_t*)w.order, (const uint32_t*)w.toff, (const uint32_t*)w.nmulti, nb, buckets, task_partial, 0, (uint64_t)0, (const uint32_t*)nullptr);
            }
            {
                ZkProfScope ps(ctx, "msm_combine");
                hipLaunchKernelGGL(k_msm_combine_wave, dim3(combine_grid((max_tasks + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)w.nmulti, (const uint32_t*)w.toff, task_partial);
                hipLaunchKernelGGL(k_msm_combine_small, dim3(combine_grid((nb + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)w.nmulti, (const uint32_t*)w.order,
                                   (const uint32_t*)w.ntasks, (const uint32_t*)w.toff, (const G1Xyzz29*)task_partial, buckets);
                hipLaunchKernelGGL(k_msm_combine, dim3(256), dim3(256), 0, ctx->stream, (const uint32_t*)w.nmulti, (const uint32_t*)w.order,
                                   (const uint32_t*)w.ntasks, (const uint32_t*)w.toff, (const G1Xyzz29*)task_partial, buckets);
                ZK_CHECK_LAUNCH(ctx);
            }
            ZK_HIP(ctx, hipEventRecord(ctx->ev_p1[par], ctx->stream));
            ZK_HIP(ctx, hipStreamWaitEvent(side, ctx->ev_p1[par], 0));
            {   // weighted bucket sum on a side stream: hides under the next MSMs
                ZkProfScope ps(ctx, "msm_reduce", side);
                if (it + 2 >= count) {
                    // the caller waits for the reductions of the last two MSMs of a batch (and of a lone one): one launch with a
                    // scalar multiplication per lane has the shorter dependent chain (about 60 additions against 150 over the
                    // levels below), at twice the work
                    const uint32_t rb = ((nb + RED_G_WIDE - 1) / RED_G_WIDE + RED_THREADS - 1) / RED_THREADS;      // <= nb / 2048 + 1 partials: fits red_pts
                    hipLaunchKernelGGL((k_msm_reduce<RED_G_WIDE>), dim3(rb, 1), dim3(RED_THREADS), 0, side, (const G1Xyzz29*)buckets, nb, partial);
                    hipLaunchKernelGGL(k_msm_window_sum, dim3(1), dim3(RED_THREADS), 0, side, (const G1Xyzz29*)partial, rb, wsum_all + it);
                    ZK_CHECK_LAUNCH(ctx);
                } else {
                    PK_TRY_MSM(wsum_enqueue(ctx, side, buckets, nb, partial, wsum_all + it));
                }
            }
            ZK_HIP(ctx, hipEventRecord(ctx->ev_p2[par], side));
        }
        if (it + grp < count) {
            if (sort_ahead) {
                // the next column: its staging (upload) is fenced into the sort stream, its sort goes into the other workspace copy,
                // free once the accumulation + combination of MSM it - 1 (recorded as ev_p1 of that iteration) are through
                ctx->stream = sort_st;
                for (; staged < it + 2; ++staged) { int rc = stage(stage_user, staged); if (rc) return rc; }
                if (it >= 1) ZK_HIP(ctx, hipStreamWaitEvent(sort_st, ctx->ev_p1[(it - 1) % 3], 0));
                PK_TRY_MSM(enqueue_sort(it + 1, (int)((it + 1) & 1), sort_st));
                ZK_HIP(ctx, hipEventRecord(ctx->ev_sorted[(it + 1) & 1], sort_st));
            } else if (stage) {
                // the columns of the NEXT step start crossing the link now, under this step's accumulation
                ctx->stream = mains[(stepno + 1) % (size_t)npipe];
                const size_t upto = it + grp + group_of(it + grp);
                for (; staged < upto; ++staged) { int rc = stage(stage_user, staged); if (rc) return rc; }
            }
        }
        it += grp;
    }
    ctx->stream = mains[0];
    if (npipe == 2) {
        ZK_HIP(ctx, hipEventRecord(ctx->ev_pipe, mains[1]));
        ZK_HIP(ctx, hipStreamWaitEvent(mains[0], ctx->ev_pipe, 0));
    }
    ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_p2[0], 0));
    if (stepno > 1) ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_p2[1], 0));
    if (stepno > 2) ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_p2[2], 0));
    PK_TRY_MSM(enqueue_tails());
    std::vector<G1Xyzz> hw(count), ht(tail_rows ? count : 0);
    if (tail_rows) ZK_HIP(ctx, hipMemcpyAsync(ht.data(), tails_dev, sizeof(G1Xyzz) * count, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipMemcpyAsync(hw.data(), wsum_all, sizeof(G1Xyzz) * count, hipMemcpyDeviceToHost, ctx->stream));
    const auto t_enq = std::chrono::steady_clock::now();
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (getenv("ZK_MSM_TRACE")) {
        const auto t_done = std::chrono::steady_clock::now();
        fprintf(stderr, "[zk msm] batch of %zu x 2^%.0f: host enqueue %.3f ms, device drained %.3f ms later\n", count, log2((double)n),
                std::chrono::duration<double, std::milli>(t_enq - t_batch0).count(), std::chrono::duration<double, std::milli>(t_done - t_enq).count());
    }
    for (size_t it = 0; it < count; ++it) {
        if (tail_rows && narrow[it] == 1) host::msm_tail2(hw.data() + it, ht.data() + it, h_out + it);
        else host::msm_tail(hw.data() + it, 1, pl.c, h_out + it);
    }
    return ZK_OK;
}
// Per-window table of an SRS basis (plan make_plan(2^k): c = k - 4 <= 16), for the columns that fill
// few windows; built on first use like the merged one.
static int srs_window_table_narrow(zk_ctx* ctx, const zk_srs* srs, int basis, size_t n, const G1Affine** out, MsmPlan* plan) {
    *out = nullptr;
    zk_srs* s = const_cast<zk_srs*>(srs);
    const uint64_t ns = 1ull << s->k;
    const MsmPlan pl = make_plan(ns);
    *plan = pl;
    const size_t bytes = sizeof(G1Affine) * ns * pl.W;
    const char* env = getenv("ZK_MSM_TABLE_GB");
    const double cap_gb = env ? atof(env) : 32.0;
    if (n < 1024 || (double)bytes > cap_gb * (double)(1ull << 30)) return ZK_OK;
    if (!s->tabn[basis]) {
        const G1Affine* rp = nullptr;
        int rc = srs_bases_rp(ctx, srs, basis, &rp);
        if (rc) return rc;
        if (hipMalloc(&s->tabn[basis], bytes) != hipSuccess) { (void)hipGetLastError(); s->tabn[basis] = nullptr; return ZK_OK; }
        hipLaunchKernelGGL(k_build_window_tables, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, ctx->stream, rp, ns, pl.c, pl.W, s->tabn[basis], 0);
        ZK_CHECK_LAUNCH(ctx);
    }
    *out = s->tabn[basis];
    return ZK_OK;
}
// commitments over an SRS basis: the merged-window path when the basis has (or can get) its window
// table -- columns flagged `narrow` take the per-window path over their own table --, the plain
// per-window path otherwise.  ZK_MSM_NARROW=0 / 1 overrides the flags (measurement knob).
int msm_batch_srs(zk_ctx* ctx, const zk_srs* srs, int basis, const Fr* const* d_scalar_ptrs, size_t count, size_t n, G1Affine* h_out, MsmStageFn stage, void* stage_user, const uint8_t* narrow) {
    // columns hinted as run-structured and resident on the device: by their run ends (runs.hip); what is left takes the paths below
    static thread_local bool runs_tried = false;     // set while the columns msm_runs_try left undone are re-submitted: hint 2 then only selects the sliced sort
    if (narrow && !stage && count && !runs_tried) {
        bool any2 = false;
        for (size_t i = 0; i < count; ++i) any2 |= narrow[i] == 2;
        if (any2) {
            std::vector<uint8_t> done(count);
            int rc = msm_runs_try(ctx, srs, basis, d_scalar_ptrs, count, n, narrow, h_out, done.data());
            if (rc) return rc;
            std::vector<size_t> rest;
            for (size_t i = 0; i < count; ++i) if (!done[i]) rest.push_back(i);
            if (rest.size() < count) {
                if (rest.empty()) return ZK_OK;
                std::vector<const Fr*> ptrs(rest.size());
                std::vector<uint8_t> hints(rest.size());
                std::vector<G1Affine> outs(rest.size());
                for (size_t j = 0; j < rest.size(); ++j) { ptrs[j] = d_scalar_ptrs[rest[j]]; hints[j] = narrow[rest[j]]; }
                {
                    struct Tried { Tried() { runs_tried = true; } ~Tried() { runs_tried = false; } } tried;
                    rc = msm_batch_srs(ctx, srs, basis, ptrs.data(), rest.size(), n, outs.data(), nullptr, nullptr, hints.data());
                }
                if (rc) return rc;
                for (size_t j = 0; j < rest.size(); ++j) h_out[rest[j]] = outs[j];
                return ZK_OK;
            }
        }
    }
    // columns hinted as sums with mostly equal increments (hint 3: a lookup's running sum): through their first differences
    // against the prefix basis (runs.hip); what is left takes the paths below as dense columns
    if (narrow && !stage && count && basis == 1) {
        bool any3 = false;
        for (size_t i = 0; i < count; ++i) any3 |= narrow[i] == 3;
        if (any3) {
            std::vector<uint8_t> done(count);
            int rc3 = msm_diff_try(ctx, srs, basis, d_scalar_ptrs, count, n, narrow, h_out, done.data());
            if (rc3) return rc3;
            std::vector<size_t> rest;
            for (size_t i = 0; i < count; ++i) if (!done[i]) rest.push_back(i);
            if (rest.empty()) return ZK_OK;
            std::vector<const Fr*> ptrs(rest.size());
            std::vector<uint8_t> hints(rest.size());
            std::vector<G1Affine> outs(rest.size());
            for (size_t j = 0; j < rest.size(); ++j) { ptrs[j] = d_scalar_ptrs[rest[j]]; hints[j] = narrow[rest[j]] == 3 ? 0 : narrow[rest[j]]; }
            rc3 = msm_batch_srs(ctx, srs, basis, ptrs.data(), rest.size(), n, outs.data(), nullptr, nullptr, hints.data());
            if (rc3) return rc3;
            for (size_t j = 0; j < rest.size(); ++j) h_out[rest[j]] = outs[j];
            return ZK_OK;
        }
    }
    const G1Affine* b = basis == 2 ? srs->pfx[1] : (basis ? srs->g_lagrange : srs->g);
    const G1Affine* brp = nullptr;
    int rc = srs_bases_rp(ctx, srs, basis, &brp);
    if (rc) return rc;
    const G1Affine* tab = nullptr;
    size_t stride = 0;
    rc = srs_window_table(ctx, srs, basis, n, &tab, &stride);
    if (rc) return rc;
    if (!tab) return msm_batch_tab(ctx, d_scalar_ptrs, count, b, brp, nullptr, 0, n, h_out, stage, stage_user);
    std::vector<uint8_t> forced;
    if (const char* e = getenv("ZK_MSM_NARROW")) { forced.assign(count, (uint8_t)(atoi(e) != 0)); narrow = forced.data(); }
    bool any = false;
    if (narrow) for (size_t i = 0; i < count; ++i) any |= narrow[i] == 1;
    const G1Affine* tabn = nullptr;
    MsmPlan pln{};
    if (any) { rc = srs_window_table_narrow(ctx, srs, basis, n, &tabn, &pln); if (rc) return rc; }
    return msm_batch_merged(ctx, d_scalar_ptrs, count, tab, stride, make_plan_merged(srs->k), n, h_out, stage, stage_user, tabn, &pln, narrow);
}
int msm_batch_rp(zk_ctx* ctx, const Fr* const* d_scalar_ptrs, size_t count, const G1Affine* d_bases, const G1Affine* d_bases_rp, size_t n, G1Affine* h_out) {
    return msm_batch_tab(ctx, d_scalar_ptrs, count, d_bases, d_bases_rp, nullptr, 0, n, h_out);
}
// Window table of an SRS basis (table[w][i] = 2^(c w) * P_i for the merged-window plan of the SRS's
// own k), built on first use and cached on the zk_srs; *out stays nullptr when the table would be
// too large (beyond ZK_MSM_TABLE_GB, default 32 GiB per basis) or the MSM is tiny.
int srs_window_table(zk_ctx* ctx, const zk_srs* srs, int basis, size_t n, const G1Affine** out, size_t* stride) {
    *out = nullptr;
    *stride = 0;
    zk_srs* s = const_cast<zk_srs*>(srs);
    const MsmPlan pl = make_plan_merged(s->k);
    const uint64_t ns = 1ull << s->k;
    const size_t bytes = sizeof(G1Affine) * ns * pl.W;
    const char* env = getenv("ZK_MSM_TABLE_GB");
    const double cap_gb = env ? atof(env) : 32.0;
    if (n < 64 || (double)bytes > cap_gb * (double)(1ull << 30) || (uint64_t)pl.W * ns >= (1ull << 31)) return ZK_OK;
    if (s->tab[basis] && s->tab_c[basis] != pl.c) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(s->tab[basis]); s->tab[basis] = nullptr; }   // plan changed (measurement knob)
    if (!s->tab[basis]) {
        const G1Affine* rp = nullptr;
        int rc = srs_bases_rp(ctx, srs, basis, &rp);
        if (rc) return rc;
        if (hipMalloc(&s->tab[basis], bytes) != hipSuccess) { (void)hipGetLastError(); s->tab[basis] = nullptr; return ZK_OK; }   // no memory: per-window path
        s->tab_c[basis] = pl.c;
        hipLaunchKernelGGL(k_build_window_tables, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, ctx->stream, rp, ns, pl.c, pl.W, s->tab[basis], pl.top_shift);
        ZK_CHECK_LAUNCH(ctx);
    }
    *out = s->tab[basis];
    *stride = ns;
    return ZK_OK;
}

int msm_run_rp(zk_ctx* ctx, const Fr* d_scalars, const G1Affine* d_bases, const G1Affine* d_bases_rp, size_t n, G1Affine* h_out) {
    return msm_batch_rp(ctx, &d_scalars, 1, d_bases, d_bases_rp, n, h_out);
}
int msm_run(zk_ctx* ctx, const Fr* d_scalars, const G1Affine* d_bases, size_t n, G1Affine* h_out) {
    return msm_run_rp(ctx, d_scalars, d_bases, nullptr, n, h_out);
}

// R'-form copy of an SRS basis, built on first use and cached on the zk_srs
int srs_bases_rp(zk_ctx* ctx, const zk_srs* srs, int basis, const G1Affine** out) {
    zk_srs* s = const_cast<zk_srs*>(srs);
    if (basis == 2) {                   // the prefix basis is kept in R' form (runs.hip builds it)
        if (!s->pfx[1]) return ctx->fail(ZK_ERR_INVALID_ARG, "the prefix basis has not been built");
        *out = s->pfx[1];
        return ZK_OK;
    }
    G1Affine** slot = basis ? &s->g_lagrange_rp : &s->g_rp;
    const G1Affine* src = basis ? s->g_lagrange : s->g;
    if (!*slot) {
        const uint64_t n = 1ull << s->k;
        ZK_HIP(ctx, hipMalloc(slot, sizeof(G1Affine) * n));
        hipLaunchKernelGGL(k_bases_to_rprime, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, src, *slot, n);
        ZK_CHECK_LAUNCH(ctx);
    }
    *out = *slot;
    return ZK_OK;
}

}  // namespace zk

// Host only: the merged-window plan of an SRS of 2^k points -- window bits, windows, and the shift the top window's digit is
// scaled by (MsmPlan::top_shift).  For the CPU tests, which check the invariant the recoding relies on: the scaled top digit of
// every canonical scalar, carry included, stays at or below 2^(c-1).
extern "C" int zk_host_msm_plan(uint32_t k, int* window_bits, int* windows, int* top_shift) {
    if (!window_bits || !windows || !top_shift || k > 28) return ZK_ERR_INVALID_ARG;
    const zk::MsmPlan m = zk::make_plan_merged(k);
    *window_bits = m.c;
    *windows = m.W;
    *top_shift = m.top_shift;
    return ZK_OK;
}

// window size / window count the commitments over this SRS use for an MSM of n points (introspection for benches)
extern "C" int zk_msm_plan(const zk_srs* srs, size_t n, int* window_bits, int* windows) {
    if (!srs || !window_bits || !windows) return ZK_ERR_INVALID_ARG;
    const uint64_t ns = 1ull << srs->k;
    const zk::MsmPlan m = zk::make_plan_merged(srs->k);
    const char* env = getenv("ZK_MSM_TABLE_GB");
    const double cap_gb = env ? atof(env) : 32.0;
    const bool merged = n >= 64 && (double)(sizeof(zk::G1Affine) * ns * m.W) <= cap_gb * (double)(1ull << 30) && (uint64_t)m.W * ns < (1ull << 31);
    const zk::MsmPlan pl = merged ? m : zk::make_plan(n);
    *window_bits = pl.c;
    *windows = pl.W;
    return ZK_OK;
}
