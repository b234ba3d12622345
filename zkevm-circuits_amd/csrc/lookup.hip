// Lookup multiplicities m(X) of the logUp ("mv-lookup") argument on the device.
//
// Reference behaviour (halo2, Scroll fork, plonk/mv_lookup/prover.rs `Argument::prepare`, SURVEY
// Appendix B.6): after the table expressions and every input tuple of the argument are
// theta-compressed, m[i] counts how many usable input rows -- over ALL input tuples -- carry the
// value of table row i; when a value occurs in several table rows the LAST one takes the whole
// count (the upstream code collects value -> row into a BTreeMap, so later rows overwrite earlier
// ones; any choice yields a valid proof); an input that is not in the table is a prover error
// (Error::ConstraintSystemFailure).
//
// Device form: an open-addressing hash table over the usable table rows (slot = highest row index
// with that value, settled with atomicCAS / atomicMax), then one probe sequence per input row
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
    // Rows are taken from the top down: the workgroups that start first carry the highest rows, so when a value repeats over a
    // long stretch (the default row a fixed table is padded with) its owner is settled by the first few waves and every later
    // wave finds a higher row in the slot with one load -- no atomic at all.  (Bottom-up, each of the 16 000 waves of a 2^20-row
    // table raised the same slot with an atomicMax: 0.26 ms per table.)
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    const bool active = idx < rows;
    const uint32_t row = active ? rows - 1u - idx : 0u;
    const Fr key = active ? ldg(table + row) : Fr::zero();
    // Fixed tables are padded with long runs of one default row.  Equal values inside a wave are
    // represented by their first lane (the highest row) only, so a run of a million equal rows
    // sends 1/64 of the probes to that value's slot instead of hammering one L2 line.
    bool rep = false;
    uint64_t todo = __ballot(active);
    while (todo) {
        const int src = (int)__builtin_ctzll(todo);
        uint32_t diff = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) diff |= key.l[i] ^ __shfl(key.l[i], src);
        const uint64_t same = __ballot(diff == 0) & todo;
        if ((int)lane == (int)__builtin_ctzll(same)) rep = true;
        todo &= ~same;
    }
    if (!rep) return;
    uint32_t h = fr_hash(key) & mask;
    for (;;) {
        // look before touching the slot atomically: a value that is already owned by a higher row
        // costs one load and no atomic
        uint32_t old = __hip_atomic_load(&slots[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == LK_EMPTY) old = atomicCAS(&slots[h], LK_EMPTY, row);
        if (old == LK_EMPTY) return;
        if (fr_same(ldg(table + old), key)) { if (row > old) atomicMax(&slots[h], row); return; }   // duplicate value: highest row owns it (LK_EMPTY never reappears: rows < 2^31)
        h = (h + 1) & mask;
    }
}
constexpr int LK_COUNT_THREADS = 1024;
constexpr int LK_BLOCK_SLOT_BITS = 11;
constexpr uint32_t LK_BLOCK_SLOTS = 1u << LK_BLOCK_SLOT_BITS;     // 2 x LK_COUNT_THREADS
// status[0] = lowest input row whose value is not in the table (LK_EMPTY if none)
__global__ void __launch_bounds__(LK_COUNT_THREADS) k_lk_count(const Fr* __restrict__ inputs, const Fr* __restrict__ table, uint32_t rows, const uint32_t* __restrict__ slots, uint32_t mask,
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
    // Disabled rows all look up the same default value, and a million atomics on one counter would
    // serialise.  Two levels of combining: lanes of a wave with the same owner share one update,
    // and the waves of the workgroup merge their updates in a small LDS hash table (owner ->
    // count) that is flushed with one global atomic per distinct owner of the workgroup.
    __shared__ uint32_t t_key[LK_BLOCK_SLOTS], t_cnt[LK_BLOCK_SLOTS];
    for (uint32_t i = threadIdx.x; i < LK_BLOCK_SLOTS; i += blockDim.x) { t_key[i] = LK_EMPTY; t_cnt[i] = 0u; }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t todo = __ballot(owner != LK_EMPTY);
    while (todo) {
        const uint32_t lead = __shfl(owner, (int)__builtin_ctzll(todo));
        const uint64_t same = __ballot(owner == lead) & todo;
        if (lane == (uint32_t)__builtin_ctzll(todo)) {
            uint32_t h = (lead * 0x9E3779B1u) >> (32 - LK_BLOCK_SLOT_BITS);
            for (;;) {       // at most blockDim.x distinct owners for 2 * blockDim.x slots: always terminates
                const uint32_t old = atomicCAS(&t_key[h], LK_EMPTY, lead);
                if (old == LK_EMPTY || old == lead) { atomicAdd(&t_cnt[h], (uint32_t)__popcll(same)); break; }
                h = (h + 1) & (LK_BLOCK_SLOTS - 1);
            }
        }
        todo &= ~same;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < LK_BLOCK_SLOTS; i += blockDim.x)
        if (t_key[i] != LK_EMPTY) atomicAdd(&counts[t_key[i]], t_cnt[i]);
}
// m[i] = counts[i] as a Montgomery residue for i < rows, 0 for rows <= i < n (halo2 leaves the unusable rows of m at zero)
__global__ void __launch_bounds__(256) k_lk_to_fr(const uint32_t* __restrict__ counts, uint32_t rows, Fr* __restrict__ m, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = Fr::zero();
    if (i < rows) { v.l[0] = counts[i]; v = to_mont(v); }
    stg(m + i, v);
}


// Enqueues the whole computation on the context's stream; d_status (one u32 on the device, set to
// 0xFFFFFFFF by the caller) receives the lowest offending input row.  No synchronisation: the
// prover runs every lookup of a proof back to back and reads all status words at once.
// d_inputs: `num_inputs` theta-compressed input vectors looked up in the same table (mv-lookup
// arguments carry one or more input tuples).
// reuse_hash: the previous call on this context was for the SAME table values and row count (consecutive lookup arguments into
// one table -- what chunk_lookups() leaves when a table has more inputs than one argument's degree allows): its hash is still in
// the scratch, only the counters start over.
int lookup_multiplicities_enqueue(zk_ctx* ctx, const Fr* const* d_inputs, size_t num_inputs, const Fr* d_table, size_t usable_rows, Fr* d_m, size_t n, uint32_t* d_status, bool reuse_hash) {
    uint32_t cap = 16;
    while (cap < 2 * usable_rows) cap <<= 1;
    // scratch: slots[cap] | counts[usable_rows]   (reused by the next enqueue: same stream, so ordered)
    uint32_t* ws = (uint32_t*)ctx->get_scratch(SC_TMP, ((size_t)cap + usable_rows + 4) * 4);
    if (!ws) return ZK_ERR_OOM;
    uint32_t *slots = ws, *counts = ws + cap;
    if (!reuse_hash) ZK_HIP(ctx, hipMemsetAsync(slots, 0xFF, (size_t)cap * 4, ctx->stream));
    ZK_HIP(ctx, hipMemsetAsync(counts, 0, (size_t)usable_rows * 4, ctx->stream));
    const uint32_t rows = (uint32_t)usable_rows;
    if (rows) {
        if (!reuse_hash) hipLaunchKernelGGL(k_lk_insert, dim3((rows + 255) / 256), dim3(256), 0, ctx->stream, d_table, rows, slots, cap - 1);
        for (size_t i = 0; i < num_inputs; ++i)
            hipLaunchKernelGGL(k_lk_count, dim3((rows + LK_COUNT_THREADS - 1) / LK_COUNT_THREADS), dim3(LK_COUNT_THREADS), 0, ctx->stream, d_inputs[i], d_table, rows,
                               (const uint32_t*)slots, cap - 1, counts, d_status);
        ZK_CHECK_LAUNCH(ctx);
    }
    hipLaunchKernelGGL(k_lk_to_fr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)counts, rows, d_m, (uint64_t)n);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}


// ---- row checks of zk_mock_verify (prover.hip): halo2 dev::MockProver::verify_at_rows_par restated for the device -----------
// Every kernel appends {kind, index, sub, row} records to `out` (first `cap` of them; *counter keeps counting).

// selected rows (row_ids, or 0 .. count - 1) whose value is not zero: VerifyFailure::ConstraintNotSatisfied
__global__ void __launch_bounds__(256) k_mock_nonzero(const Fr* __restrict__ vals, const uint32_t* __restrict__ row_ids, uint32_t count, uint32_t kind, uint32_t index, uint32_t sub,
                                                      MockFail* __restrict__ out, uint32_t cap, uint32_t* __restrict__ counter) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t row = row_ids ? row_ids[t] : t;
    if (fr_same(ldg(vals + row), Fr::zero())) return;
    const uint32_t pos = atomicAdd(counter, 1u);
    if (pos < cap) out[pos] = MockFail{kind, index, sub, row};
}
// selected rows whose (compressed) input is in none of the table rows the hash was built from: VerifyFailure::Lookup
__global__ void __launch_bounds__(256) k_mock_probe(const Fr* __restrict__ inputs, const Fr* __restrict__ table, const uint32_t* __restrict__ slots, uint32_t mask,
                                                    const uint32_t* __restrict__ row_ids, uint32_t count, uint32_t kind, uint32_t index, uint32_t sub,
                                                    MockFail* __restrict__ out, uint32_t cap, uint32_t* __restrict__ counter) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t row = row_ids ? row_ids[t] : t;
    const Fr key = ldg(inputs + row);
    uint32_t h = fr_hash(key) & mask;
    for (;;) {
        const uint32_t owner = slots[h];
        if (owner == LK_EMPTY) break;
        if (fr_same(ldg(table + owner), key)) return;
        h = (h + 1) & mask;
    }
    const uint32_t pos = atomicAdd(counter, 1u);
    if (pos < cap) out[pos] = MockFail{kind, index, sub, row};
}
// Copy constraints from the key's sigma columns alone: sigma_j[i] = delta^j' w^i' names the cell (j', i') that cell (j, i) must
// equal; `ids` holds delta^j w^i for every cell (cell j n + i) and `slots` the hash over it, so one probe inverts sigma.
// VerifyFailure::Permutation {column j, row i} where the two values differ (sub 0) or sigma names no cell at all (sub 1).
__global__ void __launch_bounds__(256) k_mock_perm(const Fr* const* __restrict__ sigma, const Fr* const* __restrict__ cols, const Fr* __restrict__ ids,
                                                   const uint32_t* __restrict__ slots, uint32_t mask, uint32_t cells, uint32_t k, uint32_t kind,
                                                   MockFail* __restrict__ out, uint32_t cap, uint32_t* __restrict__ counter) {
    const uint32_t cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= cells) return;
    const uint32_t j = cell >> k, i = cell & ((1u << k) - 1u);
    const Fr key = ldg(sigma[j] + i);
    uint32_t h = fr_hash(key) & mask, owner;
    for (;;) {
        owner = slots[h];
        if (owner == LK_EMPTY || fr_same(ldg(ids + owner), key)) break;
        h = (h + 1) & mask;
    }
    uint32_t sub = 1;
    if (owner != LK_EMPTY) {
        if (owner == cell) return;
        if (fr_same(ldg(cols[j] + i), ldg(cols[owner >> k] + (owner & ((1u << k) - 1u))))) return;
        sub = 0;
    }
    const uint32_t pos = atomicAdd(counter, 1u);
    if (pos < cap) out[pos] = MockFail{kind, j, sub, i};
}

// open-addressing hash over table[0 .. rows) in scratch `slot` (2 x rows slots, rounded up to a power of two); enqueued, no sync
int mock_hash_build(zk_ctx* ctx, const Fr* d_table, size_t rows, int scratch_slot, const uint32_t** slots_out, uint32_t* mask_out) {
    if (rows > ((size_t)1 << 30)) return ctx->fail(ZK_ERR_UNSUPPORTED, "mock verify: %zu rows exceed the 2^30 a device hash holds", rows);
    size_t cap = 16;
    while (cap < 2 * rows) cap <<= 1;
    uint32_t* slots = (uint32_t*)ctx->get_scratch(scratch_slot, cap * 4);
    if (!slots) return ZK_ERR_OOM;
    ZK_HIP(ctx, hipMemsetAsync(slots, 0xFF, cap * 4, ctx->stream));
    if (rows) hipLaunchKernelGGL(k_lk_insert, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, d_table, (uint32_t)rows, slots, (uint32_t)(cap - 1));
    ZK_CHECK_LAUNCH(ctx);
    *slots_out = slots;
    *mask_out = (uint32_t)(cap - 1);
    return ZK_OK;
}
int mock_nonzero_enqueue(zk_ctx* ctx, const Fr* d_vals, const uint32_t* d_row_ids, uint32_t count, uint32_t kind, uint32_t index, uint32_t sub, MockFail* d_out, uint32_t cap, uint32_t* d_counter) {
    if (!count) return ZK_OK;
    hipLaunchKernelGGL(k_mock_nonzero, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, d_vals, d_row_ids, count, kind, index, sub, d_out, cap, d_counter);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}
int mock_probe_enqueue(zk_ctx* ctx, const Fr* d_inputs, const Fr* d_table, const uint32_t* d_slots, uint32_t mask, const uint32_t* d_row_ids, uint32_t count,
                       uint32_t kind, uint32_t index, uint32_t sub, MockFail* d_out, uint32_t cap, uint32_t* d_counter) {
    if (!count) return ZK_OK;
    hipLaunchKernelGGL(k_mock_probe, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, d_inputs, d_table, d_slots, mask, d_row_ids, count, kind, index, sub, d_out, cap, d_counter);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}
int mock_perm_enqueue(zk_ctx* ctx, const Fr* const* d_sigma, const Fr* const* d_cols, const Fr* d_ids, const uint32_t* d_slots, uint32_t mask, uint32_t num_cols, uint32_t k,
                      uint32_t kind, MockFail* d_out, uint32_t cap, uint32_t* d_counter) {
    const uint32_t cells = num_cols << k;
    if (!cells) return ZK_OK;
    hipLaunchKernelGGL(k_mock_perm, dim3((cells + 255) / 256), dim3(256), 0, ctx->stream, d_sigma, d_cols, d_ids, d_slots, mask, cells, k, kind, d_out, cap, d_counter);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}

}  // namespace zk

using namespace zk;

extern "C" int zk_lookup_multiplicities(zk_ctx* ctx, const void* d_inputs, const void* d_table, size_t usable_rows, void* d_m, size_t n, uint64_t* bad_row) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_inputs && d_table && d_m && bad_row, "null pointer");
    ZK_REQUIRE(ctx, usable_rows <= n && n < (1ull << 31), "row counts out of range");
    *bad_row = UINT64_MAX;
    uint32_t* status = (uint32_t*)ctx->get_scratch(SC_TMP2, 64);
    if (!status) return ZK_ERR_OOM;
    ZK_HIP(ctx, hipMemsetAsync(status, 0xFF, 4, ctx->stream));
    const Fr* one_input = (const Fr*)d_inputs;
    int rc = lookup_multiplicities_enqueue(ctx, &one_input, 1, (const Fr*)d_table, usable_rows, (Fr*)d_m, n, status, false);
    if (rc) return rc;
    uint32_t st = LK_EMPTY;
    ZK_HIP(ctx, hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st != LK_EMPTY) *bad_row = st;
    return ZK_OK;
}
