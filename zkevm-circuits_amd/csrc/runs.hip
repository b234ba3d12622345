// Commitments to run-structured columns: halo2_proofs' commit_lagrange applied to permutation grand products and other
// piecewise-constant columns (reference call sites A1-A4 reach it through create_proof's
// permutation::Argument::commit, external crate).
//
// A permutation product Z stays constant over every row whose cells take part in no copy constraint; in a zkEVM-style circuit
// that is nearly every row (the SuperCircuit wires its sub-circuits with lookups, copies are few), so Z is a few hundred runs of
// equal values over 2^20 rows.  The bucket method is at its worst there -- every row of a run lands in the same bucket of every
// window -- while the sum itself collapses by Abel summation over the run ends e_0 < e_1 < ... (z constant on (e_{r-1}, e_r]):
//
//      sum_i z_i L_i  =  sum_r (z_{e_r} - z_{e_r + 1}) * P_{e_r},        P_e = sum_{i <= e} L_i,   z_n := 0.
//
// P is a prefix-sum table of the basis (64 B per point, built once per SRS and basis on first use, R' form like every base the
// MSM kernels read); a column then costs one sweep that compares neighbouring rows and collects (difference, P_e) pairs, and an
// MSM over as many points as the column has runs.  A column with more than n / 16 runs is not run-structured: it takes the
// ordinary path.  Results are the same group elements either way (exact arithmetic; the affine output is unique).
#include "ctx.hpp"
#include "ec29.hip.hpp"
#include "host_fq.hpp"

namespace zk {

constexpr int PFX_SEG = 64;               // points a thread sums one after the other
constexpr int RUN_COLS = 32;              // columns per launch of the collection kernel
struct RunCols { const Fr* p[RUN_COLS]; };

// ---- prefix-sum table of a basis -----------------------------------------------------------------------------------------
// three levels of segments of PFX_SEG points: segment totals, totals of PFX_SEG segments, a serial scan of those (n / 4096
// values), then the offsets come back down and every thread re-adds its segment, converting each prefix to affine form
__global__ void __launch_bounds__(256) k_pfx_totals_affine(const G1Affine* __restrict__ pts, uint64_t n, G1Xyzz29* __restrict__ totals) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, i0 = s * PFX_SEG;
    if (i0 >= n) return;
    G1Xyzz29 acc = identity29();
    const uint64_t i1 = min(n, i0 + PFX_SEG);
    for (uint64_t i = i0; i < i1; ++i) acc = madd29(acc, load_affine29(pts + i));
    stg29(totals + s, acc);
}
__global__ void __launch_bounds__(256) k_pfx_totals_xyzz(const G1Xyzz29* __restrict__ in, uint64_t n, G1Xyzz29* __restrict__ totals) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, i0 = s * PFX_SEG;
    if (i0 >= n) return;
    G1Xyzz29 acc = identity29();
    const uint64_t i1 = min(n, i0 + PFX_SEG);
    for (uint64_t i = i0; i < i1; ++i) acc = add29pt(acc, ldg29(in + i));
    stg29(totals + s, acc);
}
// in place: totals[i] <- sum of totals[j], j < i (one thread; n / 4096 values)
__global__ void k_pfx_serial_exclusive(G1Xyzz29* __restrict__ v, uint64_t n) {
    if (blockIdx.x || threadIdx.x) return;
    G1Xyzz29 acc = identity29();
    for (uint64_t i = 0; i < n; ++i) {
        const G1Xyzz29 cur = ldg29(v + i);
        stg29(v + i, acc);
        acc = add29pt(acc, cur);
    }
}
// in place: every segment of `child` becomes its exclusive prefix inside the parent, lifted by the parent's offset
__global__ void __launch_bounds__(256) k_pfx_down(G1Xyzz29* __restrict__ child, uint64_t n_child, const G1Xyzz29* __restrict__ parent_off) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, i0 = s * PFX_SEG;
    if (i0 >= n_child) return;
    G1Xyzz29 acc = ldg29(parent_off + s);
    const uint64_t i1 = min(n_child, i0 + PFX_SEG);
    for (uint64_t i = i0; i < i1; ++i) {
        const G1Xyzz29 cur = ldg29(child + i);
        stg29(child + i, acc);
        acc = add29pt(acc, cur);
    }
}
__global__ void __launch_bounds__(256) k_pfx_finish(const G1Affine* __restrict__ pts, uint64_t n, const G1Xyzz29* __restrict__ seg_off, G1Affine* __restrict__ out) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, i0 = s * PFX_SEG;
    if (i0 >= n) return;
    G1Xyzz29 acc = ldg29(seg_off + s);
    const uint64_t i1 = min(n, i0 + PFX_SEG);
    for (uint64_t i = i0; i < i1; ++i) {
        acc = madd29(acc, load_affine29(pts + i));
        stg(out + i, to_affine_rp(acc));
    }
}

// -(sum of v[0 .. n)) as one affine point in R' form (one thread; n / 4096 values)
__global__ void k_pfx_neg_total(const G1Xyzz29* __restrict__ v, uint64_t n, G1Affine* __restrict__ out) {
    if (blockIdx.x || threadIdx.x) return;
    G1Xyzz29 acc = identity29();
    for (uint64_t i = 0; i < n; ++i) acc = add29pt(acc, ldg29(v + i));
    G1Affine a = to_affine_rp(acc);
    if (!a.is_identity()) a.y = neg(a.y);
    stg(out, a);
}
static int srs_prefix_table(zk_ctx* ctx, const zk_srs* srs, int basis, const G1Affine** out) {
    *out = nullptr;
    zk_srs* s = const_cast<zk_srs*>(srs);
    if (s->pfx[basis]) { *out = s->pfx[basis]; return ZK_OK; }
    const uint64_t n = 1ull << s->k;
    const G1Affine* rp = nullptr;
    int rc = srs_bases_rp(ctx, srs, basis, &rp);
    if (rc) return rc;
    const uint64_t n1 = (n + PFX_SEG - 1) / PFX_SEG, n2 = (n1 + PFX_SEG - 1) / PFX_SEG;
    G1Xyzz29* t1 = (G1Xyzz29*)ctx->pool_get((n1 + n2) * sizeof(G1Xyzz29));
    if (!t1) return ZK_OK;                                                                    // no memory: the ordinary path
    G1Xyzz29* t2 = t1 + n1;
    G1Affine* tab = nullptr;
    if (hipMalloc(&tab, sizeof(G1Affine) * n) != hipSuccess) { (void)hipGetLastError(); ctx->pool_put(t1, (n1 + n2) * sizeof(G1Xyzz29)); return ZK_OK; }
    auto grid = [](uint64_t items) { return dim3((unsigned)((items + 255) / 256)); };
    hipLaunchKernelGGL(k_pfx_totals_affine, grid(n1), dim3(256), 0, ctx->stream, rp, n, t1);
    hipLaunchKernelGGL(k_pfx_totals_xyzz, grid(n2), dim3(256), 0, ctx->stream, (const G1Xyzz29*)t1, n1, t2);
    hipLaunchKernelGGL(k_pfx_serial_exclusive, dim3(1), dim3(64), 0, ctx->stream, t2, n2);
    hipLaunchKernelGGL(k_pfx_down, grid(n2), dim3(256), 0, ctx->stream, t1, n1, (const G1Xyzz29*)t2);
    hipLaunchKernelGGL(k_pfx_finish, grid(n1), dim3(256), 0, ctx->stream, rp, n, (const G1Xyzz29*)t1, tab);
    // -(P_0 + ... + P_{n-2}): what a column committed through its first differences owes for the constant part of its increments
    G1Affine* negtot = nullptr;
    if (hipMalloc(&negtot, sizeof(G1Affine)) != hipSuccess) { (void)hipGetLastError(); negtot = nullptr; }
    if (negtot) {
        hipLaunchKernelGGL(k_pfx_totals_affine, grid(n1), dim3(256), 0, ctx->stream, (const G1Affine*)tab, n - 1, t1);
        hipLaunchKernelGGL(k_pfx_totals_xyzz, grid(n2), dim3(256), 0, ctx->stream, (const G1Xyzz29*)t1, (n - 1 + PFX_SEG - 1) / PFX_SEG, t2);
        hipLaunchKernelGGL(k_pfx_neg_total, dim3(1), dim3(64), 0, ctx->stream, (const G1Xyzz29*)t2, ((n - 1 + PFX_SEG - 1) / PFX_SEG + PFX_SEG - 1) / PFX_SEG, negtot);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    ctx->pool_put(t1, (n1 + n2) * sizeof(G1Xyzz29));
    if (e != hipSuccess) { (void)hipFree(tab); if (negtot) (void)hipFree(negtot); return ctx->fail(ZK_ERR_HIP, "prefix table of the basis: %s", hipGetErrorString(e)); }
    s->pfx[basis] = tab;
    s->pfx_negtot[basis] = negtot;
    *out = tab;
    return ZK_OK;
}

// ---- run ends of a column ------------------------------------------------------------------------------------------------
// row i is a run end when z_i != z_{i+1} (z_n = 0): (z_i - z_{i+1}, P_i) goes to the column's list, in no particular order;
// a workgroup reserves room for its ends with one atomic.  counts[col] may pass `cap`: nothing is written beyond it.
// pfx == nullptr: count only (the table does not exist yet: it is built once a column shows that it will be used).
__global__ void __launch_bounds__(256) k_runs_collect(RunCols cols, uint64_t n, const G1Affine* __restrict__ pfx, Fr* __restrict__ scal, G1Affine* __restrict__ base, uint32_t cap,
                                                      uint32_t* __restrict__ counts) {
    __shared__ uint32_t wcnt[4], wbase[4];
    const Fr* __restrict__ z = cols.p[blockIdx.y];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    Fr cur = Fr::zero(), nxt = Fr::zero();
    if (i < n) cur = ldg(z + i);
    if (i + 1 < n) nxt = ldg(z + i + 1);
    bool end = false;
    if (i < n) {
#pragma unroll
        for (int q = 0; q < 8; ++q) end |= cur.l[q] != nxt.l[q];
    }
    const uint64_t bal = __ballot(end);
    if (lane == 0) wcnt[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        const uint32_t start = total ? atomicAdd(&counts[blockIdx.y], total) : 0u;
        wbase[0] = start; wbase[1] = start + wcnt[0]; wbase[2] = wbase[1] + wcnt[1]; wbase[3] = wbase[2] + wcnt[2];
    }
    __syncthreads();
    if (end) {
        const uint32_t pos = wbase[wave] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (pos < cap && pfx) {
            stg(scal + (uint64_t)blockIdx.y * cap + pos, cur - nxt);
            stg(base + (uint64_t)blockIdx.y * cap + pos, ldg(pfx + i));
        }
    }
}

// ---- the sum over the run ends, directly -----------------------------------------------------------------------------------
// A few hundred (difference, P_e) pairs per column are too few for the bucket method (a chain of some twenty-five launches per
// column, 0.4 ms each): one lane per pair multiplies by double-and-add (254 doublings + ~127 mixed additions, 1.3 ms of latency
// whatever the count), all columns of the launch side by side, then a tree sum per workgroup and one per column.
constexpr uint32_t RUNS_DIRECT_MAX = 4096;       // run ends per column up to which the direct form is used
__global__ void __launch_bounds__(256) k_runs_mul(const Fr* __restrict__ scal, const G1Affine* __restrict__ base, uint32_t cap, const uint32_t* __restrict__ counts, G1Xyzz29* __restrict__ partial) {
    __shared__ G1Xyzz29 sh[256];
    const uint32_t col = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cnt = counts[col];             // beyond cap the list is incomplete: such a column is not summed here
    G1Xyzz29 acc = identity29();
    if (cnt <= cap && cnt <= RUNS_DIRECT_MAX && j < cnt) {
        const Fr k = from_mont(ldg(scal + (uint64_t)col * cap + j));
        const G1Affine29 p = load_affine29(base + (uint64_t)col * cap + j);
        int top = 255;
        while (top >= 0 && !((k.l[top >> 5] >> (top & 31)) & 1u)) --top;
#pragma unroll 1
        for (int bit = top; bit >= 0; --bit) {
            acc = dbl29pt(acc);
            if ((k.l[bit >> 5] >> (bit & 31)) & 1u) acc = madd29(acc, p);
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = add29pt(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) stg29(partial + (uint64_t)col * gridDim.x + blockIdx.x, sh[0]);
}
__global__ void __launch_bounds__(64) k_runs_sum(const G1Xyzz29* __restrict__ partial, uint32_t per_col, G1Xyzz* __restrict__ out) {
    const uint32_t col = blockIdx.x;
    if (threadIdx.x) return;
    G1Xyzz29 acc = identity29();
    for (uint32_t i = 0; i < per_col; ++i) acc = add29pt(acc, ldg29(partial + (uint64_t)col * per_col + i));
    stg(out + col, to_std_xyzz(acc));
}

// ---- a scalar times ONE fixed point -------------------------------------------------------------------------------------------
// c * Q for the fixed point Q = -(P_0 + ... + P_{n-2}) of a basis (msm_diff_try below): a table of digit * 2^(8 w) * Q over the 32
// byte windows of a scalar (512 KiB, built once per SRS and basis: one lane per entry, double-and-add) turns the 254-step chain of a
// single lane into 32 table reads and a five-level sum inside one wave.
constexpr int FIXED_WINDOWS = 32, FIXED_DIGITS = 256;
__global__ void __launch_bounds__(256) k_fixed_table(const G1Affine* __restrict__ q, G1Affine* __restrict__ tab) {
    const uint32_t w = blockIdx.x, d = threadIdx.x;           // entry [w][d] = (d << 8 w) * Q
    const G1Affine29 p = load_affine29(q);
    G1Xyzz29 acc = identity29();
    if (d) {
#pragma unroll 1
        for (int bit = 7; bit >= 0; --bit) {
            acc = dbl29pt(acc);
            if ((d >> bit) & 1u) acc = madd29(acc, p);
        }
#pragma unroll 1
        for (uint32_t t = 0; t < 8 * w; ++t) acc = dbl29pt(acc);
    }
    stg(tab + (size_t)w * FIXED_DIGITS + d, to_affine_rp(acc));
}
// out[col] = c[col] * Q from the table; one wave per column, lane w < 32 holds window w
__global__ void __launch_bounds__(64) k_fixed_mul(const Fr* __restrict__ c, const G1Affine* __restrict__ tab, G1Xyzz* __restrict__ out) {
    __shared__ G1Xyzz29 sh[64];
    const uint32_t col = blockIdx.x, w = threadIdx.x;
    G1Xyzz29 acc = identity29();
    if (w < (uint32_t)FIXED_WINDOWS) {
        const Fr k = from_mont(ldg(c + col));
        const uint32_t digit = (k.l[w >> 2] >> (8 * (w & 3))) & 0xFFu;
        if (digit) acc = madd29(acc, load_affine29(tab + (size_t)w * FIXED_DIGITS + digit));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 16; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = add29pt(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) stg(out + col, to_std_xyzz(sh[0]));
}
static int srs_fixed_table(zk_ctx* ctx, const zk_srs* srs, int basis, const G1Affine** out) {
    *out = nullptr;
    zk_srs* s = const_cast<zk_srs*>(srs);
    if (s->pfx_negtot_tab[basis]) { *out = s->pfx_negtot_tab[basis]; return ZK_OK; }
    if (!s->pfx_negtot[basis]) return ZK_OK;
    G1Affine* tab = nullptr;
    if (hipMalloc(&tab, sizeof(G1Affine) * FIXED_WINDOWS * FIXED_DIGITS) != hipSuccess) { (void)hipGetLastError(); return ZK_OK; }
    hipLaunchKernelGGL(k_fixed_table, dim3(FIXED_WINDOWS), dim3(FIXED_DIGITS), 0, ctx->stream, (const G1Affine*)s->pfx_negtot[basis], tab);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(tab); return ctx->fail(ZK_ERR_HIP, "fixed-base table: %s", hipGetErrorString(e)); }
    s->pfx_negtot_tab[basis] = tab;
    *out = tab;
    return ZK_OK;
}

// Commits the columns hinted as run-structured (narrow[i] == 2) that do have few runs; done[i] = 1 for those, the others are left
// to the caller's ordinary path.  No-op (all zero) when the feature is off, the columns are small, or memory is short.
int msm_runs_try(zk_ctx* ctx, const zk_srs* srs, int basis, const Fr* const* d_scalar_ptrs, size_t count, size_t n, const uint8_t* narrow, G1Affine* h_out, uint8_t* done) {
    memset(done, 0, count);
    if (!narrow || n < 4096) return ZK_OK;
    if (const char* e = getenv("ZK_MSM_RUNS")) if (atoi(e) == 0) return ZK_OK;
    std::vector<size_t> sel;
    for (size_t i = 0; i < count; ++i) if (narrow[i] == 2) sel.push_back(i);
    if (sel.empty()) return ZK_OK;
    const uint32_t cap = (uint32_t)std::min<size_t>(n / 16, (size_t)1 << 16);
    int rc = ZK_OK;
    if (!srs->pfx[basis]) {
        // The prefix table costs 64 B per SRS point and an affine conversion per point: before it exists, one counting sweep
        // decides whether any of these columns will use it (an aggregation circuit's permutation products have a run per row)
        uint32_t* d_counts = (uint32_t*)ctx->pool_get(256);
        if (!d_counts) return ZK_OK;
        std::vector<size_t> keep;
        for (size_t first = 0; first < sel.size(); first += RUN_COLS) {
            const size_t cnt = std::min<size_t>(RUN_COLS, sel.size() - first);
            RunCols rcols{};
            for (size_t j = 0; j < cnt; ++j) rcols.p[j] = d_scalar_ptrs[sel[first + j]];
            uint32_t h_counts[RUN_COLS];
            hipError_t e = hipMemsetAsync(d_counts, 0, 256, ctx->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_runs_collect, dim3((unsigned)((n + 255) / 256), (unsigned)cnt), dim3(256), 0, ctx->stream, rcols, (uint64_t)n, (const G1Affine*)nullptr, (Fr*)nullptr, (G1Affine*)nullptr, cap, d_counts);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(h_counts, d_counts, cnt * 4, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { ctx->pool_put(d_counts, 256); return ctx->fail(ZK_ERR_HIP, "run count: %s", hipGetErrorString(e)); }
            for (size_t j = 0; j < cnt; ++j) if (h_counts[j] <= cap) keep.push_back(sel[first + j]);
        }
        ctx->pool_put(d_counts, 256);
        if (keep.empty()) return ZK_OK;
        sel.swap(keep);
    }
    const G1Affine* pfx = nullptr;
    rc = srs_prefix_table(ctx, srs, basis, &pfx);
    if (rc) return rc;
    if (!pfx) return ZK_OK;
    for (size_t first = 0; first < sel.size(); first += RUN_COLS) {
        const size_t cnt = std::min<size_t>(RUN_COLS, sel.size() - first);
        const size_t bytes = cnt * cap * (sizeof(Fr) + sizeof(G1Affine)) + 256;
        char* buf = (char*)ctx->pool_get(bytes);
        if (!buf) return ZK_OK;
        uint32_t* d_counts = (uint32_t*)buf;
        Fr* d_scal = (Fr*)(buf + 256);
        G1Affine* d_base = (G1Affine*)(buf + 256 + cnt * cap * sizeof(Fr));
        auto release = [&](int code) { ctx->pool_put(buf, bytes); return code; };
        RunCols rcols{};
        for (size_t j = 0; j < cnt; ++j) rcols.p[j] = d_scalar_ptrs[sel[first + j]];
        hipError_t e = hipMemsetAsync(d_counts, 0, 256, ctx->stream);
        if (e != hipSuccess) return release(ctx->fail(ZK_ERR_HIP, "run collection: %s", hipGetErrorString(e)));
        hipLaunchKernelGGL(k_runs_collect, dim3((unsigned)((n + 255) / 256), (unsigned)cnt), dim3(256), 0, ctx->stream, rcols, (uint64_t)n, pfx, d_scal, d_base, cap, d_counts);
        uint32_t h_counts[RUN_COLS];
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h_counts, d_counts, cnt * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return release(ctx->fail(ZK_ERR_HIP, "run collection: %s", hipGetErrorString(e)));
        // columns with at most RUNS_DIRECT_MAX run ends: one launch for all of them
        uint32_t direct_max = 0;
        auto direct = [&](size_t j) { return h_counts[j] <= cap && h_counts[j] <= RUNS_DIRECT_MAX; };
        for (size_t j = 0; j < cnt; ++j) if (direct(j)) direct_max = std::max(direct_max, h_counts[j]);
        if (direct_max) {
            const uint32_t blocks = (direct_max + 255) / 256;
            const size_t pbytes = (size_t)cnt * blocks * sizeof(G1Xyzz29) + cnt * sizeof(G1Xyzz);
            char* pb = (char*)ctx->pool_get(pbytes);
            if (pb) {
                G1Xyzz29* d_part = (G1Xyzz29*)pb;
                G1Xyzz* d_res = (G1Xyzz*)(pb + (size_t)cnt * blocks * sizeof(G1Xyzz29));
                hipLaunchKernelGGL(k_runs_mul, dim3(blocks, (unsigned)cnt), dim3(256), 0, ctx->stream, (const Fr*)d_scal, (const G1Affine*)d_base, cap, (const uint32_t*)d_counts, d_part);
                hipLaunchKernelGGL(k_runs_sum, dim3((unsigned)cnt), dim3(64), 0, ctx->stream, (const G1Xyzz29*)d_part, blocks, d_res);
                std::vector<G1Xyzz> hres(cnt);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipMemcpyAsync(hres.data(), d_res, cnt * sizeof(G1Xyzz), hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                ctx->pool_put(pb, pbytes);
                if (e != hipSuccess) return release(ctx->fail(ZK_ERR_HIP, "run sums: %s", hipGetErrorString(e)));
                for (size_t j = 0; j < cnt; ++j) {
                    if (!direct(j)) continue;
                    host::msm_tail(hres.data() + j, 1, 1, h_out + sel[first + j]);
                    done[sel[first + j]] = 1;
                }
            }
        }
        for (size_t j = 0; j < cnt; ++j) {
            const size_t col = sel[first + j];
            if (done[col] || h_counts[j] > cap) continue;              // done above / not run-structured after all: the ordinary path
            if (h_counts[j] == 0) { memset(h_out + col, 0, sizeof(G1Affine)); done[col] = 1; continue; }      // the zero column
            const Fr* sp = d_scal + j * cap;
            const G1Affine* bp = d_base + j * cap;
            rc = msm_batch_tab(ctx, &sp, 1, bp, bp, nullptr, 0, h_counts[j], h_out + col);
            if (rc) return release(rc);
            done[col] = 1;
        }
        ctx->pool_put(buf, bytes);
    }
    return ZK_OK;
}

// ---- columns committed through their first differences ------------------------------------------------------------------------
// A lookup's running sum phi (logUp: phi_{i+1} = phi_i + 1 / (f_i + beta) - m_i / (t_i + beta)) grows by the SAME amount on every
// row where the lookup is switched off and the table row is unused -- in a circuit whose lookups are conditional, most rows.
// With d_j = phi_{j+1} - phi_j and c the most frequent increment,
//
//      sum_i phi_i L_i  =  phi_{n-1} P_{n-1} - sum_{j <= n-2} d_j P_j  =  MSM_P(s) - c * (P_0 + ... + P_{n-2}),
//      s_j = c - d_j (j <= n - 2),  s_{n-1} = phi_{n-1}:
//
// s is zero wherever the increment is the common one, so the commitment becomes an MSM of a mostly-zero column over the prefix
// basis P (window tables of its own, built on first use: basis index 2 of msm_batch_srs, where the column is judged like any
// other -- a column whose increments are not mostly equal simply takes the dense path there) plus one scalar multiplication of a
// fixed point.  Only full-length columns (n = 2^k) of at least 4096 rows; ZK_MSM_DIFF=0 turns it off.
constexpr int DIFF_SAMPLES = 256;
__global__ void __launch_bounds__(DIFF_SAMPLES) k_diff_mode(RunCols cols, uint64_t n, Fr* __restrict__ c_out, uint32_t* __restrict__ votes_out) {
    __shared__ Fr d[DIFF_SAMPLES];
    __shared__ uint32_t votes[DIFF_SAMPLES];
    const Fr* __restrict__ z = cols.p[blockIdx.x];
    const uint64_t j = (uint64_t)threadIdx.x * ((n - 1) / DIFF_SAMPLES) + (threadIdx.x * 7 + blockIdx.x) % ((n - 1) / DIFF_SAMPLES);      // n >= 4096: j <= n - 2
    const Fr mine = ldg(z + j + 1) - ldg(z + j);
    d[threadIdx.x] = mine;
    __syncthreads();
    uint32_t v = 0;
    for (int t = 0; t < DIFF_SAMPLES; ++t) {
        bool eq = true;
#pragma unroll
        for (int q = 0; q < 8; ++q) eq &= d[t].l[q] == mine.l[q];
        v += eq;
    }
    votes[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        for (int t = 1; t < DIFF_SAMPLES; ++t) if (votes[t] > votes[best]) best = t;
        stg(c_out + blockIdx.x, votes[best] * 2 >= DIFF_SAMPLES ? d[best] : Fr::zero());      // no majority: c = 0 (s = -d, a dense column)
        votes_out[blockIdx.x] = votes[best];
    }
}
__global__ void __launch_bounds__(256) k_diff_sparse(RunCols cols, uint64_t n, const Fr* __restrict__ c_in, Fr* __restrict__ s_out) {
    const Fr* __restrict__ z = cols.p[blockIdx.y];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr cur = ldg(z + i);
    Fr r = cur;                                      // s_{n-1} = phi_{n-1}
    if (i + 1 < n) r = ldg(c_in + blockIdx.y) - (ldg(z + i + 1) - cur);
    stg(s_out + (uint64_t)blockIdx.y * n + i, r);
}

int msm_diff_try(zk_ctx* ctx, const zk_srs* srs, int basis, const Fr* const* d_scalar_ptrs, size_t count, size_t n, const uint8_t* narrow, G1Affine* h_out, uint8_t* done) {
    memset(done, 0, count);
    if (!narrow || basis != 1 || n < 4096 || n != ((size_t)1 << srs->k)) return ZK_OK;
    if (const char* e = getenv("ZK_MSM_DIFF")) if (atoi(e) == 0) return ZK_OK;
    std::vector<size_t> sel;
    for (size_t i = 0; i < count; ++i) if (narrow[i] == 3) sel.push_back(i);
    if (sel.empty()) return ZK_OK;
    constexpr size_t CHUNK = 16;                       // columns whose difference images exist at a time (16 x n x 32 B)
    int rc = ZK_OK;
    {   // Only sums whose increments ARE mostly equal go this way (three quarters of the sampled increments agree): the others would
        // be dense columns over the prefix basis too -- same cost, but the basis' window tables would be built for nothing.
        char* vb = (char*)ctx->pool_get(RUN_COLS * (sizeof(Fr) + 4));
        if (!vb) return ZK_OK;
        Fr* d_c0 = (Fr*)vb;
        uint32_t* d_votes = (uint32_t*)(vb + RUN_COLS * sizeof(Fr));
        std::vector<size_t> keep;
        for (size_t first = 0; first < sel.size(); first += RUN_COLS) {
            const size_t cnt = std::min<size_t>(RUN_COLS, sel.size() - first);
            RunCols rcols{};
            for (size_t j = 0; j < cnt; ++j) rcols.p[j] = d_scalar_ptrs[sel[first + j]];
            uint32_t h_votes[RUN_COLS];
            hipLaunchKernelGGL(k_diff_mode, dim3((unsigned)cnt), dim3(DIFF_SAMPLES), 0, ctx->stream, rcols, (uint64_t)n, d_c0, d_votes);
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(h_votes, d_votes, cnt * 4, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { ctx->pool_put(vb, RUN_COLS * (sizeof(Fr) + 4)); return ctx->fail(ZK_ERR_HIP, "increment sampling: %s", hipGetErrorString(e)); }
            for (size_t j = 0; j < cnt; ++j) if (h_votes[j] * 4 >= (uint32_t)DIFF_SAMPLES * 3) keep.push_back(sel[first + j]);
        }
        ctx->pool_put(vb, RUN_COLS * (sizeof(Fr) + 4));
        if (keep.empty()) return ZK_OK;
        sel.swap(keep);
    }
    const G1Affine* pfx = nullptr;
    rc = srs_prefix_table(ctx, srs, basis, &pfx);
    if (rc) return rc;
    if (!pfx || !srs->pfx_negtot[basis]) return ZK_OK;
    const G1Affine* fixed_tab = nullptr;
    rc = srs_fixed_table(ctx, srs, basis, &fixed_tab);
    if (rc) return rc;
    if (!fixed_tab) return ZK_OK;
    for (size_t first = 0; first < sel.size(); first += CHUNK) {
        const size_t cnt = std::min(CHUNK, sel.size() - first);
        const size_t s_bytes = cnt * n * sizeof(Fr), aux_bytes = CHUNK * (sizeof(Fr) + sizeof(G1Xyzz) + 4) + 256;
        Fr* d_s = (Fr*)ctx->pool_get(s_bytes);
        char* aux = (char*)ctx->pool_get(aux_bytes);
        if (!d_s || !aux) { ctx->pool_put(d_s, s_bytes); ctx->pool_put(aux, aux_bytes); return ZK_OK; }
        // error paths: kernels that read d_s / aux may still be in flight on the stream -- drain it before the blocks go back to the pool
        auto release = [&](int code) { (void)hipStreamSynchronize(ctx->stream); ctx->pool_put(d_s, s_bytes); ctx->pool_put(aux, aux_bytes); return code; };
        Fr* d_c = (Fr*)aux;
        G1Xyzz* d_res = (G1Xyzz*)(d_c + CHUNK);
        uint32_t* d_votes = (uint32_t*)(d_res + CHUNK);
        RunCols rcols{};
        for (size_t j = 0; j < cnt; ++j) rcols.p[j] = d_scalar_ptrs[sel[first + j]];
        hipLaunchKernelGGL(k_diff_mode, dim3((unsigned)cnt), dim3(DIFF_SAMPLES), 0, ctx->stream, rcols, (uint64_t)n, d_c, d_votes);
        hipLaunchKernelGGL(k_diff_sparse, dim3((unsigned)((n + 255) / 256), (unsigned)cnt), dim3(256), 0, ctx->stream, rcols, (uint64_t)n, (const Fr*)d_c, d_s);
        // c * (-(P_0 + ... + P_{n-2})) for every column of the chunk, out of the point's fixed-base table; the results are fetched
        // after the batch below has drained the stream (no synchronisation of their own)
        hipLaunchKernelGGL(k_fixed_mul, dim3((unsigned)cnt), dim3(64), 0, ctx->stream, (const Fr*)d_c, fixed_tab, d_res);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return release(ctx->fail(ZK_ERR_HIP, "difference commitments: %s", hipGetErrorString(e)));
        // the mostly-zero images over the prefix basis, judged and committed like any batch of columns
        std::vector<const void*> sp(cnt);
        for (size_t j = 0; j < cnt; ++j) sp[j] = d_s + j * n;
        std::vector<uint8_t> kind(cnt);
        rc = sample_narrow_dev(ctx, sp.data(), cnt, n, kind.data());
        if (rc) return release(rc);
        std::vector<G1Affine> part(cnt);
        rc = msm_batch_srs(ctx, srs, 2, (const Fr* const*)sp.data(), cnt, n, part.data(), nullptr, nullptr, kind.data());
        if (rc) return release(rc);
        std::vector<G1Xyzz> hres(cnt);
        e = hipMemcpyAsync(hres.data(), d_res, cnt * sizeof(G1Xyzz), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return release(ctx->fail(ZK_ERR_HIP, "difference commitments: %s", hipGetErrorString(e)));
        for (size_t j = 0; j < cnt; ++j) {
            G1Xyzz a = G1Xyzz::identity();
            if (!part[j].is_identity()) { a.x = part[j].x; a.y = part[j].y; a.zz = Fq::one(); a.zzz = Fq::one(); }
            host::msm_tail2(&a, hres.data() + j, h_out + sel[first + j]);
            done[sel[first + j]] = 1;
        }
        ctx->pool_put(d_s, s_bytes);
        ctx->pool_put(aux, aux_bytes);
    }
    return ZK_OK;
}

}  // namespace zk
