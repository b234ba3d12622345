// Quotient-polynomial evaluation over the extended domain: halo2_proofs
// plonk::evaluation::{Evaluator::evaluate_h, GraphEvaluator} + vanishing divide_by_vanishing_poly
// (external crate; reached from create_proof, reference call sites A1-A3).
//
// The host flattens every constraint of the circuit -- custom gates, permutation and lookup
// identities alike -- into one postfix program over *columns in extended-coset evaluation form*
// (halo2's `Calculation` / `ValueSource` list, restated as a stack machine):
//
//      PUSH_COL c, rot     push column c at row (i + rot * 2^(ext_k - k)) mod 2^ext_k
//      PUSH_CONST j        push consts[j]              (constants, challenges, beta/gamma/theta)
//      ADD SUB MUL NEG SQUARE DOUBLE MUL_CONST j ADD_CONST j
//      FOLD j              acc = acc * consts[j] + pop()     (the `acc * y + term` folding)
//      TEE_TMP t           tmp[t] = top (stack unchanged)    -- halo2's GraphEvaluator keeps every shared
//      PUSH_TMP t          push tmp[t]                          sub-expression as an intermediate; a value used by
//                                                               several gates is computed once and parked here
//      END
//
// One lane per extended-domain row; control flow is uniform across the grid (every lane runs the
// same instruction), so there is no divergence.  The two topmost operands live in registers, deeper
// ones in LDS laid out [slot][limb][lane] (4-byte strided: bank-conflict-free), column reads are fully coalesced for
// rot = 0 and shifted-coalesced for rot != 0.  The result is multiplied by the precomputed
// 1/(X^n - 1) on the coset (period 2^(ext_k-k)) before it is written.
// HBM side: 32 B per (column, rotation) read + 32 B written per row -- the streaming-bound member
// of the path.
//
// Arithmetic: the stack machine computes on nine 29-bit limbs (ff29.cuh) -- the carry-free
// Montgomery product runs 1.6x faster than the 8 x 32 CIOS one.  Stack values stay in halo2curves'
// R = 2^256 Montgomery form, normalised and below 2p.  mul29 divides by R' = 2^261, so a product of
// two R-form values multiplies one operand by 32 first (a 5-bit limb shift: 64p still fits the
// 261-bit container and 2p * 64p < 2^261 p); constants that only ever multiply (MUL_CONST, FOLD,
// the vanishing inverses) are tabulated in R' form instead and need no shift.  Spilled stack
// slots are packed to 8 words, so the LDS footprint is that of the 32-bit representation.
#include "ctx.hpp"
#include "ff29.cuh"

namespace zk {

using Q29 = F29<Fr29P>;

// limb idx of K*p, normalised (limbs 0..7 < 2^29)
template <class P>
__host__ __device__ constexpr uint32_t kp_norm(int K, int idx) {
    uint64_t carry = 0;
    uint32_t out = 0;
    for (int i = 0; i <= idx; ++i) {
        const uint64_t v = (uint64_t)K * P::M(i) + carry;
        out = i < 8 ? (uint32_t)(v & MASK29) : (uint32_t)v;
        carry = v >> 29;
    }
    return out;
}
// r (limbs < 2^31, value < 4p) -> normalised representative below 2p: carry-propagate, then subtract
// 2p with a signed borrow chain and keep the difference unless it went negative
__device__ __forceinline__ Q29 q_settle(Q29 r) {
    normalize29(r);
    Q29 d;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int32_t s_ = (int32_t)r.l[i] - (int32_t)kp_norm<Fr29P>(2, i) + c;
        d.l[i] = i < 8 ? ((uint32_t)s_ & MASK29) : (uint32_t)s_;
        c = s_ >> 29;
    }
    const bool neg_ = (int32_t)d.l[8] < 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = neg_ ? r.l[i] : d.l[i];
    return r;
}
__device__ __forceinline__ Q29 q_add(const Q29& a, const Q29& b) { return q_settle(add29(a, b)); }          // a, b < 2p
__device__ __forceinline__ Q29 q_sub(const Q29& a, const Q29& b) { return q_settle(sub29k<2>(a, b)); }      // a - b + 2p in (0, 4p)
// x * 32 for a normalised x < 2p: limbs stay normalised, value < 64p < 2^260
__device__ __forceinline__ Q29 q_shl5(const Q29& x) {
    Q29 r;
    r.l[0] = (x.l[0] << 5) & MASK29;
#pragma unroll
    for (int i = 1; i < 8; ++i) r.l[i] = ((x.l[i] << 5) & MASK29) | (x.l[i - 1] >> 24);
    r.l[8] = (x.l[8] << 5) | (x.l[7] >> 24);
    return r;
}
__device__ __forceinline__ Q29 q_mul(const Q29& a, const Q29& b) { return mul29(a, q_shl5(b)); }            // R-form x R-form -> R-form, < 2p

enum QOp : uint32_t { Q_END = 0, Q_PUSH_COL = 1, Q_PUSH_CONST = 2, Q_ADD = 3, Q_SUB = 4, Q_MUL = 5, Q_NEG = 6, Q_SQUARE = 7, Q_DOUBLE = 8, Q_FOLD = 9, Q_MUL_CONST = 10, Q_ADD_CONST = 11, Q_TEE_TMP = 12, Q_PUSH_TMP = 13 };

constexpr int Q_THREADS = 256;
constexpr int Q_MAX_STACK = 16;
constexpr uint32_t Q_MAX_TMP = 4096;     // intermediates live in HBM, [slot][row]: 32 MiB per slot at 2^20 rows

struct QStack {
    uint32_t* base;   // [slot][limb][lane]
    __device__ __forceinline__ Fr get(int slot) const {
        Fr r;
#pragma unroll
        for (int k = 0; k < 8; ++k) r.l[k] = base[(slot * 8 + k) * Q_THREADS + threadIdx.x];
        return r;
    }
    __device__ __forceinline__ void put(int slot, const Fr& v) const {
#pragma unroll
        for (int k = 0; k < 8; ++k) base[(slot * 8 + k) * Q_THREADS + threadIdx.x] = v.l[k];
    }
};

__global__ void __launch_bounds__(Q_THREADS)
k_quotient_eval(const uint32_t* __restrict__ prog, uint32_t prog_len, const Fr* const* __restrict__ cols, const Fr* __restrict__ consts,
                const Fr* __restrict__ consts_rp /* the same constants in R' form */, const Fr* __restrict__ t_evals /* R' form */, uint32_t ext_k, uint32_t k,
                Fr* __restrict__ out, Fr* tmp /* [slot][row]: the row's intermediates, written and read by its own lane */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    QStack st{smem};
    const uint64_t ne = 1ull << ext_k;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < ne;
    const uint32_t rot_scale = 1u << (ext_k - k);
    const Q29 zero29 = unpack29<Fr29P>(Fr::zero());
    Q29 acc = zero29;
    // The two topmost stack elements live in registers (t0 = top, t1 = second); element j < sp - 2
    // lives in LDS slot j.  Runs like `a b MUL c SUB` never touch LDS, and the LDS footprint per
    // lane (what bounds occupancy here) is max_depth - 2 slots.
    Q29 t0 = zero29, t1 = zero29;
    int sp = 0;
    auto push = [&](const Q29& v) {
        if (sp >= 2) st.put(sp - 2, pack29_raw(t1));
        t1 = t0;
        t0 = v;
        ++sp;
    };
    auto drop_to = [&](const Q29& top) {      // two operands consumed, `top` is the new top of stack
        --sp;
        t0 = top;
        if (sp >= 2) t1 = unpack29<Fr29P>(st.get(sp - 2));
    };
    for (uint32_t pc = 0; pc < prog_len; ++pc) {
        const uint32_t op = prog[3 * pc], a = prog[3 * pc + 1], b = prog[3 * pc + 2];
        if (op == Q_END) break;
        switch (op) {
            case Q_PUSH_COL: {
                const int64_t rot = (int32_t)b;
                const uint64_t row = (i + (uint64_t)(rot * (int64_t)rot_scale)) & (ne - 1);
                push(unpack29<Fr29P>(live ? ldg(cols[a] + row) : Fr::zero()));
                break;
            }
            case Q_PUSH_CONST: push(unpack29<Fr29P>(ldg(consts + a))); break;
            case Q_ADD: drop_to(q_add(t1, t0)); break;
            case Q_SUB: drop_to(q_sub(t1, t0)); break;
            case Q_MUL: drop_to(q_mul(t1, t0)); break;
            case Q_NEG: t0 = q_sub(zero29, t0); break;
            case Q_SQUARE: t0 = q_mul(t0, t0); break;
            case Q_DOUBLE: t0 = q_add(t0, t0); break;
            case Q_FOLD: acc = q_add(mul29(acc, unpack29<Fr29P>(ldg(consts_rp + a))), t0); drop_to(t1); break;
            case Q_MUL_CONST: t0 = mul29(t0, unpack29<Fr29P>(ldg(consts_rp + a))); break;
            case Q_ADD_CONST: t0 = q_add(t0, unpack29<Fr29P>(ldg(consts + a))); break;
            case Q_TEE_TMP: if (live) tmp[(uint64_t)a * ne + i] = pack29_raw(t0); break;                       // normalised, < 2p < 2^256
            case Q_PUSH_TMP: push(live ? unpack29<Fr29P>(tmp[(uint64_t)a * ne + i]) : zero29); break;
            default: break;
        }
    }
    if (live) {
        if (t_evals) acc = mul29(acc, unpack29<Fr29P>(ldg(t_evals + (i & (rot_scale - 1)))));
        stg(out + i, pack29_lt2p(acc));
    }
}

// host-side validation of a program: stack discipline and operand ranges
static int validate_program(zk_ctx* ctx, const uint32_t* prog, uint32_t len, uint32_t ncols, uint32_t nconsts, int* max_depth, uint32_t* num_tmp) {
    int sp = 0, mx = 0;
    std::vector<bool> defined;
    *num_tmp = 0;
    for (uint32_t pc = 0; pc < len; ++pc) {
        const uint32_t op = prog[3 * pc], a = prog[3 * pc + 1];
        switch (op) {
            case Q_END: *max_depth = mx; return ZK_OK;
            case Q_PUSH_COL: if (a >= ncols) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient program: column %u out of range at pc %u", a, pc); ++sp; break;
            case Q_PUSH_CONST: if (a >= nconsts) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient program: constant %u out of range at pc %u", a, pc); ++sp; break;
            case Q_ADD: case Q_SUB: case Q_MUL: if (sp < 2) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient program: stack underflow at pc %u", pc); --sp; break;
            case Q_NEG: case Q_SQUARE: case Q_DOUBLE: if (sp < 1) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient program: stack underflow at pc %u", pc); break;
            case Q_MUL_CONST: case Q_ADD_CONST: if (sp < 1 || a >= nconsts) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient program: bad operand at pc %u", pc); break;
            case Q_FOLD: if (sp < 1 || a >= nconsts) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient program: bad FOLD at pc %u", pc); --sp; break;
            case Q_TEE_TMP:
                if (sp < 1 || a >= Q_MAX_TMP) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient program: bad TEE_TMP at pc %u", pc);
                if (a >= defined.size()) defined.resize(a + 1, false);
                defined[a] = true;
                if (a + 1 > *num_tmp) *num_tmp = a + 1;
                break;
            case Q_PUSH_TMP:
                if (a >= defined.size() || !defined[a]) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient program: intermediate %u read before it is written (pc %u)", a, pc);
                ++sp;
                break;
            default: return ctx->fail(ZK_ERR_INVALID_ARG, "quotient program: unknown opcode %u at pc %u", op, pc);
        }
        if (sp > mx) mx = sp;
        if (sp > Q_MAX_STACK) return ctx->fail(ZK_ERR_UNSUPPORTED, "quotient program: stack deeper than %d", Q_MAX_STACK);
    }
    *max_depth = mx;
    return ZK_OK;
}

// t_evaluations[j] = 1 / ((zeta * w_ext^j)^n - 1), j < 2^(ext_k - k)   (EvaluationDomain::new)
static void vanishing_inverses(uint32_t k, uint32_t ext_k, std::vector<Fr>* out) {
    const uint32_t cnt = 1u << (ext_k - k);
    Fr zn = fr_zeta();
    for (uint32_t i = 0; i < k; ++i) zn = sqr(zn);            // zeta^n
    Fr step = fr_root_of_unity(ext_k);
    for (uint32_t i = 0; i < k; ++i) step = sqr(step);        // w_ext^n
    out->resize(cnt);
    Fr cur = zn;
    for (uint32_t j = 0; j < cnt; ++j) {
        (*out)[j] = fr_inv_host(cur - Fr::one());
        cur = cur * step;
    }
}

}  // namespace zk

using namespace zk;

extern "C" int zk_quotient_eval(zk_ctx* ctx, const uint32_t* h_program, uint32_t num_instr, const void* const* h_col_ptrs, uint32_t num_cols,
                                const void* h_consts, uint32_t num_consts, uint32_t k, uint32_t ext_k, int divide_by_vanishing, void* d_out) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, h_program && d_out && (h_col_ptrs || !num_cols) && (h_consts || !num_consts), "null pointer");
    ZK_REQUIRE(ctx, k <= ext_k && ext_k <= 28, "need k <= ext_k <= 28");
    int depth = 0;
    uint32_t num_tmp = 0;
    int rc = validate_program(ctx, h_program, num_instr, num_cols, num_consts, &depth, &num_tmp);
    if (rc) return rc;
    Fr* d_tmp = nullptr;
    if (num_tmp) {
        d_tmp = (Fr*)ctx->get_scratch(SC_QTMP, ((size_t)num_tmp << ext_k) * sizeof(Fr));
        if (!d_tmp) return ZK_ERR_OOM;
    }
    if (depth < 1) depth = 1;
    std::vector<Fr> tev;
    if (divide_by_vanishing) vanishing_inverses(k, ext_k, &tev);
    // constants that multiply (MUL_CONST, FOLD) and the vanishing inverses go to the device in R' = 2^261 form too: x 32
    auto times32 = [](Fr x) { for (int j = 0; j < 5; ++j) x = dbl(x); return x; };
    std::vector<Fr> consts_rp((const Fr*)h_consts, (const Fr*)h_consts + num_consts);
    for (Fr& c : consts_rp) c = times32(c);
    for (Fr& t : tev) t = times32(t);
    const size_t prog_bytes = (size_t)num_instr * 12 + 12, col_bytes = (size_t)(num_cols ? num_cols : 1) * 8;
    const size_t const_bytes = (size_t)(num_consts ? num_consts : 1) * sizeof(Fr) * 2, tev_bytes = tev.size() * sizeof(Fr);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    char* d = (char*)ctx->get_scratch(SC_POLY, al(prog_bytes) + al(col_bytes) + al(const_bytes) + al(tev_bytes) + 256);
    if (!d) return ZK_ERR_OOM;
    uint32_t* d_prog = (uint32_t*)d;
    const Fr** d_cols = (const Fr**)(d + al(prog_bytes));
    Fr* d_consts = (Fr*)(d + al(prog_bytes) + al(col_bytes));
    Fr* d_tev = (Fr*)(d + al(prog_bytes) + al(col_bytes) + al(const_bytes));
    std::vector<uint32_t> prog(h_program, h_program + (size_t)num_instr * 3);
    prog.push_back(Q_END); prog.push_back(0); prog.push_back(0);
    ZK_HIP(ctx, hipMemcpyAsync(d_prog, prog.data(), prog.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    if (num_cols) ZK_HIP(ctx, hipMemcpyAsync(d_cols, h_col_ptrs, (size_t)num_cols * 8, hipMemcpyHostToDevice, ctx->stream));
    if (num_consts) ZK_HIP(ctx, hipMemcpyAsync(d_consts, h_consts, (size_t)num_consts * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    if (num_consts) ZK_HIP(ctx, hipMemcpyAsync(d_consts + num_consts, consts_rp.data(), (size_t)num_consts * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    if (!tev.empty()) ZK_HIP(ctx, hipMemcpyAsync(d_tev, tev.data(), tev_bytes, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // staging vectors are stack-owned
    const uint64_t ne = 1ull << ext_k;
    const size_t lds = (size_t)(depth > 2 ? depth - 2 : 1) * 8 * Q_THREADS * 4;    // the two topmost elements are in registers
    if (!ctx->quotient_attr_set) {
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_quotient_eval, hipFuncAttributeMaxDynamicSharedMemorySize, Q_MAX_STACK * 8 * Q_THREADS * 4));
        ctx->quotient_attr_set = true;
    }
    ZkProfScope ps(ctx, ctx->prof_tag ? ctx->prof_tag : "quotient_eval");
    hipLaunchKernelGGL(k_quotient_eval, dim3((unsigned)((ne + Q_THREADS - 1) / Q_THREADS)), dim3(Q_THREADS), lds, ctx->stream, (const uint32_t*)d_prog,
                       num_instr + 1, (const Fr* const*)d_cols, (const Fr*)d_consts, (const Fr*)(d_consts + num_consts), tev.empty() ? (const Fr*)nullptr : (const Fr*)d_tev, ext_k, k, (Fr*)d_out, d_tmp);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}

namespace zk { __global__ void k_powers(Fr base, Fr mul, Fr* out, uint32_t count, int rprime); }

extern "C" int zk_fr_powers(zk_ctx* ctx, const void* h_base, const void* h_mul, void* d_out, size_t n) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, h_base && h_mul && d_out, "null pointer");
    ZK_REQUIRE(ctx, n <= (1ull << 28), "too many powers");
    if (!n) return ZK_OK;
    hipLaunchKernelGGL(zk::k_powers, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, *(const Fr*)h_base, *(const Fr*)h_mul, (Fr*)d_out, (uint32_t)n, 0);
    ZK_CHECK_LAUNCH(ctx);
    return ZK_OK;
}
