// Lookup multiplicities m(X) of the logUp ("mv-lookup") argument on the device.
//
// Reference behaviour (halo2, Scroll fork, plonk/mv_lookup/prover.rs `Argument::prepare`, SURVEY
// Appendix B.6): after the input and table expressions are theta-compressed, m[i] counts how many
// usable input rows carry the value of table row i; when a value occurs in several table rows the
// first one takes the whole count; an input that is not in the table is a prover error.
//
// Device form: an open-addressing hash table over the usable table rows (slot = lowest row index
// with that value, settled with atomicCAS / atomicMin), then one probe sequence per input row
// (atomicAdd on the owning row's counter).  Values are canonical Montgomery residues, so equality
// of the eight limbs is equality in Fr.  2 n slots for n rows: expected probe length < 1.5.
#include "ctx.hpp"

namespace zk {

constexpr uint32_t LK_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t fr_hash(const Fr& a) {
    uint32_t h = a.l[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) h = (h ^ a.l[i]) * 0x9E3779B1u + (h >> 15);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;   // murmur3 finaliser
    return h;
}
__device__ __forceinline__ bool fr_same(const Fr& a, const Fr& b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d |= a.l[i] ^ b.l[i];
    return d == 0;
}

__global__ void __launch_bounds__(256) k_lk_insert(const Fr* __restrict__ table, uint32_t rows, uint32_t* slots, uint32_t mask) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const Fr key = ldg(table + row);
    uint32_t h = fr_hash(key) & mask;
    for (;;) {
        // fixed tables are padded with long runs of one default row: look before touching the slot
        // atomically, so a value that is already owned by a lower row costs one load and no atomic
        uint32_t old = __hip_atomic_load(&slots[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == LK_EMPTY) old = atomicCAS(&slots[h], LK_EMPTY, row);
        if (old == LK_EMPTY) return;
        if (fr_same(ldg(table + old), key)) { if (row < old) atomicMin(&slots[h], row); return; }   // duplicate value: lowest row owns it
        h = (h + 1) & mask;
    }
}
// status[0] = lowest input row whose value is not in the table (LK_EMPTY if none)
__global__ void __launch_bounds__(256) k_lk_count(const Fr* __restrict__ inputs, const Fr* __restrict__ table, uint32_t rows, const uint32_t* __restrict__ slots, uint32_t mask,
                                                  uint32_t* __restrict__ counts, uint32_t* __restrict__ status) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t owner = LK_EMPTY;
    if (row < rows) {
        const Fr key = ldg(inputs + row);
        uint32_t h = fr_hash(key) & mask;
        for (;;) {
            owner = slots[h];
            if (owner == LK_EMPTY) { atomicMin(status, row); break; }
            if (fr_same(ldg(table + owner), key)) break;
            h = (h + 1) & mask;
        }
    }
    // one atomic per distinct owner in the wave: disabled rows all look up the same default value,
    // and a million atomics on one counter would serialise
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t todo = __ballot(owner != LK_EMPTY);
    while (todo) {
        const uint32_t lead = __shfl(owner, (int)__builtin_ctzll(todo));
        const uint64_t same = __ballot(owner == lead) & todo;
        if (lane == (uint32_t)__builtin_ctzll(todo)) atomicAdd(&counts[lead], (uint32_t)__popcll(same));
        todo &= ~same;
    }
}
// m[i] = counts[i] as a Montgomery residue for i < rows, 0 for rows <= i < n (the caller overwrites the blinding rows)
__global__ void __launch_bounds__(256) k_lk_to_fr(const uint32_t* __restrict__ counts, uint32_t rows, Fr* __restrict__ m, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = Fr::zero();
    if (i < rows) { v.l[0] = counts[i]; v = to_mont(v); }
    stg(m + i, v);
}

}  // namespace zk

using namespace zk;

extern "C" int zk_lookup_multiplicities(zk_ctx* ctx, const void* d_inputs, const void* d_table, size_t usable_rows, void* d_m, size_t n, uint64_t* bad_row) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_inputs && d_table && d_m && bad_row, "null pointer");
    ZK_REQUIRE(ctx, usable_rows <= n && n < (1ull << 31), "row counts out of range");
    *bad_row = UINT64_MAX;
    uint32_t cap = 16;
    while (cap < 2 * usable_rows) cap <<= 1;
    // scratch: slots[cap] | counts[usable_rows] | status[1]
    uint32_t* ws = (uint32_t*)ctx->get_scratch(SC_TMP, ((size_t)cap + usable_rows + 4) * 4);
    if (!ws) return ZK_ERR_OOM;
    uint32_t *slots = ws, *counts = ws + cap, *status = counts + usable_rows;
    ZK_HIP(ctx, hipMemsetAsync(slots, 0xFF, (size_t)cap * 4, ctx->stream));
    ZK_HIP(ctx, hipMemsetAsync(counts, 0, (size_t)usable_rows * 4, ctx->stream));
    ZK_HIP(ctx, hipMemsetAsync(status, 0xFF, 4, ctx->stream));
    const uint32_t rows = (uint32_t)usable_rows;
    if (rows) {
        const dim3 g((rows + 255) / 256), t(256);
        hipLaunchKernelGGL(k_lk_insert, g, t, 0, ctx->stream, (const Fr*)d_table, rows, slots, cap - 1);
        hipLaunchKernelGGL(k_lk_count, g, t, 0, ctx->stream, (const Fr*)d_inputs, (const Fr*)d_table, rows, (const uint32_t*)slots, cap - 1, counts, status);
        ZK_CHECK_LAUNCH(ctx);
    }
    hipLaunchKernelGGL(k_lk_to_fr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)counts, rows, (Fr*)d_m, (uint64_t)n);
    ZK_CHECK_LAUNCH(ctx);
    uint32_t st = LK_EMPTY;
    ZK_HIP(ctx, hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st != LK_EMPTY) *bad_row = st;
    return ZK_OK;
}
