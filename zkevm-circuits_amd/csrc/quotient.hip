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
// Arithmetic: the stack machine computes on nine 29-bit limbs (ff29.hip.hpp) -- the carry-free
// Montgomery product runs 1.6x faster than the 8 x 32 CIOS one.  Stack values stay in halo2curves'
// R = 2^256 Montgomery form.  mul29 divides by R' = 2^261, so a product of two R-form values
// multiplies one operand by 32 first (a 5-bit limb shift, folded into the unpacking of a memory
// operand); constants that only ever multiply (MUL_CONST, FOLD, the vanishing inverses) are
// tabulated in R' form instead and need no shift.
//
// Lowering (host, `lower_program`): the caller's postfix program is rebuilt as expression trees and
// re-emitted for the machine the kernel really implements --
//   * every binary operation whose right (or, for ADD / MUL / SUB, left) operand is a column, a
//     parked intermediate or a constant takes that operand from memory (ADD_COL, SUB_COL, RSUB_COL,
//     MUL_COL, FOLD_COL, ADD_CONST, MUL_CONST): a gate like q * (a * b - c) runs as
//     PUSH a, MUL_COL b, SUB_COL c, MUL_COL q, FOLD with no stack traffic at all;
//   * the memory operand of instruction pc + 1 is loaded before instruction pc executes (its ~1000
//     ALU cycles cover the load), instruction words are fetched two ahead;
//   * sums are not reduced when they are produced: the lowering tracks, per stack entry, a bound on
//     the value (in multiples of p) and on the limbs (in multiples of 2^29) and sets "settle this
//     operand first" bits only where the consumer's precondition would fail (bounds below).
// Results are bit-identical to the plain evaluation: every step is exact arithmetic mod p and the
// value written is the canonical representative.
//
// Round 6: two kernels run a lowered program -- k_quotient_eval2 (the default: one stack entry in registers, in-place products, the loop
// body a flat chain of single-armed ifs over a host-made class mask) and k_quotient_eval (round 5's, ZK_QUOTIENT_KERNEL=1), bit-identical.
// A large program that is a sum of terms and reads its operands many times is cut into slices that run side by side over the same rows
// and find each other's operands in the caches (plan_slices near the end of this file).
//
// Bounds (V = value bound in units of p, L = limb bound in units of 2^29; settled = (2, 1); a column,
// constant or parked intermediate is canonical = (1, 1)):
//   mul29(a, b)    needs b normalised, limbs of a < 2^31.2 (L <= 4), a * b < 2^261 p  (V_a V_b < 168);
//                  result (2, 1).  b = 32 x (a canonical operand): V_a <= 5;  b = 32 x (settled): V_a V_b <= 5.
//   add29          (V_a + V_b, L_a + L_b), kept <= (4, 4) so that the value can always be settled
//   a - b          = a + K p (balanced limbs) - b with b normalised and below K p: (V_a + K, L_a + 2),
//                  K = 1 for canonical b, 2 for settled b
//   FOLD           acc = mul29(acc, y) + t: acc is only ever the first operand of a product by a
//                  constant, so it is never settled: L_t <= 3
#include "ctx.hpp"
#include "ff29.hip.hpp"
#include <unordered_map>

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
// r (limbs < 2^31, value < 8p) -> normalised representative below 2p: one more round for the values the relaxed bounds let grow past 4p
__device__ __forceinline__ Q29 q_settle8(Q29 r) {
    normalize29(r);
    Q29 d;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int32_t s_ = (int32_t)r.l[i] - (int32_t)kp_norm<Fr29P>(4, i) + c;
        d.l[i] = i < 8 ? ((uint32_t)s_ & MASK29) : (uint32_t)s_;
        c = s_ >> 29;
    }
    const bool neg_ = (int32_t)d.l[8] < 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = neg_ ? r.l[i] : d.l[i];
    return q_settle(r);
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

enum QOp : uint32_t { Q_END = 0, Q_PUSH_COL = 1, Q_PUSH_CONST = 2, Q_ADD = 3, Q_SUB = 4, Q_MUL = 5, Q_NEG = 6, Q_SQUARE = 7, Q_DOUBLE = 8, Q_FOLD = 9, Q_MUL_CONST = 10, Q_ADD_CONST = 11, Q_TEE_TMP = 12, Q_PUSH_TMP = 13,
                      // produced by the lowering only (never part of a caller's program): the second operand comes from memory
                      K_ADD_COL = 16, K_SUB_COL = 17, K_RSUB_COL = 18, K_MUL_COL = 19, K_FOLD_COL = 20, K_NOP = 21 };
constexpr uint32_t K_SETTLE0 = 0x100, K_SETTLE1 = 0x200;      // word 0 of a lowered instruction: settle t0 / t1 before executing
constexpr uint32_t K_NORM0 = 0x400, K_NORM1 = 0x800;          // ... or only propagate its carries (limbs back below 2^29, value unchanged): a third of a settle
constexpr uint32_t K_SETTLE0_8 = 0x1000, K_SETTLE1_8 = 0x2000; // ... or settle a value that may have reached 8p (q_settle8)
constexpr int K_CONST_SHIFT = 16;                              // K_FOLD_COL keeps its constant index in the bits above

// A program constant as the kernel reads it: the nine 29-bit limbs, unpacked on the host once per launch (the kernel used to
// spend 27 vector instructions per constant operand on it).  The index is wave-uniform, so the limbs arrive by scalar loads
// and stay in scalar registers (mul29_ub).  64-byte records.
struct QC29 { uint32_t l[16]; };
#ifndef Q_THREADS_N
#define Q_THREADS_N 256
#endif
constexpr int Q_THREADS = Q_THREADS_N;      // lanes per workgroup (no cooperation between waves: the size only sets the granularity of the LDS allocation)
// how many instructions ahead of its use a memory operand is requested (1: while the previous instruction computes; 2: one more)
#ifndef Q_PREFETCH_DIST
#define Q_PREFETCH_DIST 1
#endif
constexpr int Q_MAX_STACK = 16;
constexpr uint32_t Q_MAX_TMP = 4096;     // intermediates live in HBM, [slot][row]: 32 MiB per slot at 2^20 rows

struct QStack {
    uint32_t* base;   // [slot][limb][lane], nine raw limbs: a spilled value keeps its (possibly unsettled) form
    __device__ __forceinline__ Q29 get(int slot) const {
        Q29 r;
#pragma unroll
        for (int k = 0; k < 9; ++k) r.l[k] = base[(slot * 9 + k) * Q_THREADS + threadIdx.x];
        return r;
    }
    __device__ __forceinline__ void put(int slot, const Q29& v) const {
#pragma unroll
        for (int k = 0; k < 9; ++k) base[(slot * 9 + k) * Q_THREADS + threadIdx.x] = v.l[k];
    }
};
// limbs of 32 * a for a < 2^256 (the R -> R' step of a product, folded into the unpacking)
__device__ __forceinline__ Q29 unpack29_x32(const Fr& a) {
    Q29 r;
    r.l[0] = (a.l[0] << 5) & MASK29;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const int bit = 29 * i - 5, w = bit >> 5, sh = bit & 31;
        uint32_t v = a.l[w] >> sh;
        if (sh > 3 && w + 1 < 8) v |= a.l[w + 1] << (32 - sh);
        r.l[i] = i == 8 ? v : (v & MASK29);
    }
    return r;
}
static inline bool k_has_mem_host(uint32_t w0) {
    const uint32_t o = w0 & 0xffu;
    return o == Q_PUSH_COL || (o >= K_ADD_COL && o <= K_FOLD_COL);
}
__device__ __forceinline__ bool k_has_mem(uint32_t w0) {
    const uint32_t o = w0 & 0xffu;
    return o == Q_PUSH_COL || (o >= K_ADD_COL && o <= K_FOLD_COL);
}

// Runs a LOWERED program (lower_program below): `prog` holds prog_len instructions followed by two END triples.
// FULL: the grid covers the domain exactly (2^ext_k >= Q_THREADS), no lane needs masking.
#ifdef Q_WAVES_PER_EU
#define Q_OCC_ATTR __attribute__((amdgpu_waves_per_eu(Q_WAVES_PER_EU, Q_WAVES_PER_EU)))
#else
#define Q_OCC_ATTR
#endif
// ACC_MEM (round 6): the accumulator of the FOLD instructions lives in memory ([limb][row], raw limbs), not in registers.  A compiled
// class program folds once per factor group -- 82 FOLDs in 34 185 instructions of the EVM-style program -- yet as a loop-carried
// register value the accumulator was copied (nine v_mov) at the end of EVERY case of the interpreter's switch, whichever instruction
// ran.  Programs that fold on most instructions (linear combinations: FOLD_COL chains) keep it in registers (ACC_MEM = false).
// K_FIRST_FOLD marks the first fold of a program: the accumulator is zero there, nothing is read.
constexpr uint32_t K_FIRST_FOLD = 0x4000;
template <bool FULL, bool ACC_MEM>
__global__ void __launch_bounds__(Q_THREADS) Q_OCC_ATTR
k_quotient_eval(const uint32_t* __restrict__ prog, uint32_t prog_len, const Fr* const* __restrict__ cols, const QC29* __restrict__ consts,
                const QC29* __restrict__ consts_rp /* the same constants in R' form */, const Fr* __restrict__ t_evals /* R' form */, uint32_t ext_k, uint32_t k,
                Fr* __restrict__ out, Fr* tmp /* [slot][row]: the row's intermediates, written and read by its own lane */, uint32_t* acc_mem /* ACC_MEM: [limb][row] */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    QStack st{smem};
    const uint64_t ne = 1ull << ext_k;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;            // ext_k <= 28: row indices fit 32 bits
    const bool live = FULL || i < (uint32_t)ne;
    const uint32_t rot_scale = 1u << (ext_k - k), row_mask = (uint32_t)ne - 1u;
    const Q29 zero29 = unpack29<Fr29P>(Fr::zero());
    Q29 acc = zero29;                         // ACC_MEM: unused (dead), the folds go through acc_mem
    // The two topmost stack elements live in registers (t0 = top, t1 = second); element j < sp - 2
    // lives in LDS slot j.
    Q29 t0 = zero29, t1 = zero29;
    int sp = 0;
    auto push_shift = [&]() {                 // makes room on top: the caller writes t0 next
        if (sp >= 2) st.put(sp - 2, t1);
        t1 = t0;
        ++sp;
    };
    auto drop_to = [&](const Q29& top) {      // two operands consumed, `top` is the new top of stack
        --sp;
        t0 = top;
        if (sp >= 2) t1 = st.get(sp - 2);
    };
    // A column pointer comes out of a table in memory, so the compiler cannot tell its address space and would emit FLAT loads;
    // those count on lgkmcnt as well as vmcnt, and every wait for a scalar load (instruction words, constants) would then also wait
    // for the operand that is being prefetched.  The columns are device memory: say so (global_load, vmcnt only).
    typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
    using GU4 = const U32x4 __attribute__((address_space(1)));
    using GU1 = uint32_t __attribute__((address_space(1)));
    auto load_row = [&](uint32_t a, uint32_t row) -> Fr {
        GU4* q = (GU4*)(uintptr_t)(cols[a] + row);
        const U32x4 lo = q[0], hi = q[1];
        Fr r;
        r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
        r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
        return r;
    };
    auto load = [&](uint32_t a, uint32_t b) -> Fr {
        const uint32_t row = (i + b * rot_scale) & row_mask;          // b = the rotation as a two's complement word: wraps like the domain
        if (FULL) return load_row(a, row);
        return live ? load_row(a, row) : Fr::zero();
    };
    auto cst = [&](const QC29* __restrict__ tab, uint32_t j) -> Q29 {
        Q29 r;
#pragma unroll
        for (int q = 0; q < 9; ++q) r.l[q] = tab[j].l[q];
        return r;
    };
    auto acc_get = [&](uint32_t w0_) -> Q29 {
        if (!ACC_MEM) return acc;
        Q29 r = zero29;
        if (!(w0_ & K_FIRST_FOLD) && live) {
            GU1* q = (GU1*)(uintptr_t)(acc_mem + i);
#pragma unroll
            for (int l = 0; l < 9; ++l) r.l[l] = q[(uint64_t)l << ext_k];
        }
        return r;
    };
    auto acc_put = [&](const Q29& v) {
        if (!ACC_MEM) { acc = v; return; }
        if (live) {
            GU1* q = (GU1*)(uintptr_t)(acc_mem + i);
#pragma unroll
            for (int l = 0; l < 9; ++l) q[(uint64_t)l << ext_k] = v.l[l];
        }
    };
    // Instruction words are fetched two ahead, the memory operand one ahead: the load of instruction pc + 1 is in flight while
    // instruction pc computes.  The raw operand `m` of instruction pc is unpacked into limbs FIRST, then the same registers take the
    // load for pc + 1 (no m_cur = m_next rotation: eight register copies an iteration).
    uint32_t w0 = prog[0], w1 = prog[1], w2 = prog[2];
    uint32_t n0 = prog[3], n1 = prog[4], n2 = prog[5];
    Fr m;                                      // only read by instructions that have a memory operand, which loaded it
    if (k_has_mem(w0)) m = load(w1, w2);
    for (uint32_t pc = 0; pc < prog_len; ++pc) {
        const uint32_t f0 = prog[3 * pc + 6], f1 = prog[3 * pc + 7], f2 = prog[3 * pc + 8];
        const uint32_t op = w0 & 0xffu;
        if (op == Q_END) break;
        Q29 mq = zero29;
        if (k_has_mem(w0)) mq = op == K_MUL_COL ? unpack29_x32(m) : unpack29<Fr29P>(m);
        if (k_has_mem(n0)) m = load(n1, n2);
        if (w0 & (K_SETTLE0 | K_SETTLE1 | K_NORM0 | K_NORM1 | K_SETTLE0_8 | K_SETTLE1_8)) {        // one test on the common path (three instructions in four carry no flag)
            if (w0 & K_SETTLE0) t0 = q_settle(t0);
            if (w0 & K_SETTLE1) t1 = q_settle(t1);
            if (w0 & K_NORM0) normalize29(t0);
            if (w0 & K_NORM1) normalize29(t1);
            if (w0 & K_SETTLE0_8) t0 = q_settle8(t0);
            if (w0 & K_SETTLE1_8) t1 = q_settle8(t1);
        }
        switch (op) {
            case Q_PUSH_COL: push_shift(); t0 = mq; break;
            case Q_PUSH_CONST: push_shift(); t0 = cst(consts, w1); break;
            case Q_ADD: drop_to(add29(t1, t0)); break;
            case Q_SUB: { Q29 d = sub29k<2>(t1, t0); normalize29(d); drop_to(d); break; }     // the top limb of a settled subtrahend may borrow: carry it out before anyone multiplies
            case Q_MUL: drop_to(mul29(t1, q_shl5(t0))); break;
            case Q_NEG: t0 = sub29k<2>(zero29, t0); normalize29(t0); break;
            case Q_SQUARE: t0 = mul29(t0, q_shl5(t0)); break;
            case Q_DOUBLE: t0 = add29(t0, t0); break;
            case Q_FOLD: acc_put(add29(mul29_ub(acc_get(w0), cst(consts_rp, w1)), t0)); drop_to(t1); break;
            case Q_MUL_CONST: t0 = mul29_ub(t0, cst(consts_rp, w1)); break;
            case Q_ADD_CONST: t0 = add29(t0, cst(consts, w1)); break;
            case Q_TEE_TMP: if (live) stg(tmp + (((uint64_t)w1 << ext_k) + i), pack29_lt2p(t0)); break;             // parked canonical: read back as a column
            case K_ADD_COL: t0 = add29(t0, mq); break;
            case K_SUB_COL: t0 = sub29k<2>(t0, mq); break;                                  // canonical subtrahend: its top limb is below that of 2p, no borrow (with K = 1 the top limb of p - b could borrow)
            case K_RSUB_COL: t0 = sub29k<2>(mq, t0); normalize29(t0); break;
            case K_MUL_COL: t0 = mul29(t0, mq); break;
            case K_FOLD_COL: acc_put(add29(mul29_ub(acc_get(w0), cst(consts_rp, w0 >> K_CONST_SHIFT)), mq)); break;
            default: break;
        }
        w0 = n0; w1 = n1; w2 = n2;
        n0 = f0; n1 = f1; n2 = f2;
    }
    if (live) {
        Q29 a = acc_get(0);
        if (t_evals) stg(out + i, pack29_lt2p(mul29(a, unpack29<Fr29P>(ldg(t_evals + (i & (rot_scale - 1u)))))));
        else { normalize29(a); stg(out + i, reduce_lazy29(a)); }       // acc < 6p, never settled on the way
    }
}

// ---- the interpreter with fixed register roles (round 6, second half) ----------------------------------------------------------
// k_quotient_eval above keeps the two topmost stack entries in registers and lets every case of its switch produce NEW values for
// them: the ISA carries 18 + 18 v_mov per interpreted instruction (the phi copies of t0 and t1 at the join, whichever case ran) on top
// of the rotations t1 = t0 of a push -- 1.3 M of the 5.3 M vector instructions a wave spends on the EVM-style class program.  Here
//   * ONE stack entry lives in registers (t0); every deeper entry is in LDS (a binary stack operation reads its other operand from
//     there: nine ds_read instead of eighteen v_mov);
//   * every operation UPDATES t0 IN PLACE -- the products through asm statements whose result takes the first factor's registers
//     (mul29_ipa / mul29_ub_ipa / mul29_ipb: limb j of the factor is last read one column before result limb j is written), sums and
//     differences limb by limb -- and the loop body is a flat chain of single-armed ifs over a class mask (QClass below), so its joins have nothing to copy;
//   * the second operand B (a memory operand unpacked on arrival, or the stack entry below the top) has its own nine registers and
//     dies with the instruction;
//   * memory operands are unpacked with v_alignbit (16 / 17 instructions instead of 27).
// Same instruction set, same lowering, same bounds (header comment); bit-identical results.  ZK_QUOTIENT_KERNEL=1 runs the older kernel.
__device__ __forceinline__ void q_unpack_to(Q29& r, const Fr& a) {
    r.l[0] = a.l[0] & MASK29;
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        r.l[i] = __builtin_amdgcn_alignbit(a.l[w + 1], a.l[w], sh) & MASK29;
    }
    r.l[8] = a.l[7] >> 8;
}
__device__ __forceinline__ void q_unpack_x32_to(Q29& r, const Fr& a) {       // limbs of 32 * a
    r.l[0] = (a.l[0] << 5) & MASK29;
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        const int bit = 29 * i - 5, w = bit >> 5, sh = bit & 31;
        r.l[i] = __builtin_amdgcn_alignbit(a.l[w + 1], a.l[w], sh) & MASK29;
    }
    r.l[8] = a.l[7] >> 3;
}
__device__ __forceinline__ void q_shl5_ip(Q29& x) {                          // x <- 32 x (normalised x < 2p), top down so that no limb is read after it is written
    x.l[8] = (x.l[8] << 5) | (x.l[7] >> 24);
#pragma unroll
    for (int i = 7; i >= 1; --i) x.l[i] = ((x.l[i] << 5) & MASK29) | (x.l[i - 1] >> 24);
    x.l[0] = (x.l[0] << 5) & MASK29;
}
__device__ __forceinline__ void q_settle_ip(Q29& r) {
    normalize29(r);
    uint32_t d[9];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int32_t s_ = (int32_t)r.l[i] - (int32_t)kp_norm<Fr29P>(2, i) + c;
        d[i] = i < 8 ? ((uint32_t)s_ & MASK29) : (uint32_t)s_;
        c = s_ >> 29;
    }
    const bool neg_ = (int32_t)d[8] < 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = neg_ ? r.l[i] : d[i];
}
__device__ __forceinline__ void q_settle8_ip(Q29& r) {
    normalize29(r);
    uint32_t d[9];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int32_t s_ = (int32_t)r.l[i] - (int32_t)kp_norm<Fr29P>(4, i) + c;
        d[i] = i < 8 ? ((uint32_t)s_ & MASK29) : (uint32_t)s_;
        c = s_ >> 29;
    }
    const bool neg_ = (int32_t)d[8] < 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = neg_ ? r.l[i] : d[i];
    q_settle_ip(r);
}

// What an instruction does, as one bit per step of the interpreter's loop body (word 3 of the 4-word instructions k_quotient_eval2 reads; made by
// q_class_mask on the host).  The loop body is a flat sequence of `if (mask & bit) { step }`: every step updates its registers in place and the
// only joins are those of single-armed ifs, which the register coalescer resolves without copies -- a `switch (op)` does not survive the
// compiler (cases merged and re-split around shared tails, phi copies of the whole top-of-stack at every join), and tests on separate bits of a
// word cannot be threaded into one another the way comparisons of one opcode are.
enum QClass : uint32_t {
    C_SPILL = 1u << 0,       // make room on top: PUSH_COL, PUSH_CONST
    C_UNPACK_T = 1u << 1,    // t0 <- memory operand                       PUSH_COL
    C_UNPACK_B = 1u << 2,    // B <- memory operand                        ADD_COL SUB_COL RSUB_COL FOLD_COL
    C_UNPACK_B32 = 1u << 3,  // B <- 32 x memory operand                   MUL_COL
    C_POP_B = 1u << 4,       // B <- the entry below the top               ADD SUB MUL
    C_HASMEM = 1u << 5,      // the instruction has a memory operand (tested on the NEXT instruction: its load is issued one instruction early)
    C_FLAGS = 1u << 6,       // any settle / carry-propagation request in word 0
    C_MULV = 1u << 7,        // t0 <- t0 * B                               MUL_COL
    C_MULC = 1u << 8,        // t0 <- t0 * const                           MUL_CONST
    C_ADDV = 1u << 9,        // t0 <- t0 + B                               ADD ADD_COL
    C_SUBV = 1u << 10,       // t0 <- t0 - B                               SUB_COL
    C_FOLDC = 1u << 11,      // acc <- acc * const + B                     FOLD_COL
    C_RARE = 1u << 12,       // any of the steps below
    C_PUSHC = 1u << 13,      // t0 <- const                                PUSH_CONST
    C_ADDC = 1u << 14,       // t0 <- t0 + const                           ADD_CONST
    C_RSUB = 1u << 15,       // t0 <- B - t0                               SUB RSUB_COL
    C_MULS = 1u << 16,       // t0 <- B * t0 (stack product)               MUL
    C_NEG = 1u << 17, C_SQ = 1u << 18, C_DBL = 1u << 19, C_TEE = 1u << 20,
    C_FOLD = 1u << 21,       // acc <- acc * const + t0, pop               FOLD
};
static uint32_t q_class_mask(uint32_t w0) {
    uint32_t m = 0;
    switch (w0 & 0xffu) {
        case Q_PUSH_COL: m = C_SPILL | C_UNPACK_T | C_HASMEM; break;
        case Q_PUSH_CONST: m = C_SPILL | C_RARE | C_PUSHC; break;
        case Q_ADD: m = C_POP_B | C_ADDV; break;
        case Q_SUB: m = C_POP_B | C_RARE | C_RSUB; break;
        case Q_MUL: m = C_POP_B | C_RARE | C_MULS; break;
        case Q_NEG: m = C_RARE | C_NEG; break;
        case Q_SQUARE: m = C_RARE | C_SQ; break;
        case Q_DOUBLE: m = C_RARE | C_DBL; break;
        case Q_FOLD: m = C_RARE | C_FOLD; break;
        case Q_MUL_CONST: m = C_MULC; break;
        case Q_ADD_CONST: m = C_RARE | C_ADDC; break;
        case Q_TEE_TMP: m = C_RARE | C_TEE; break;
        case K_ADD_COL: m = C_UNPACK_B | C_HASMEM | C_ADDV; break;
        case K_SUB_COL: m = C_UNPACK_B | C_HASMEM | C_SUBV; break;
        case K_RSUB_COL: m = C_UNPACK_B | C_HASMEM | C_RARE | C_RSUB; break;
        case K_MUL_COL: m = C_UNPACK_B32 | C_HASMEM | C_MULV; break;
        case K_FOLD_COL: m = C_UNPACK_B | C_HASMEM | C_FOLDC; break;
        default: break;            // K_NOP, Q_END
    }
    if (w0 & (K_SETTLE0 | K_SETTLE1 | K_NORM0 | K_NORM1 | K_SETTLE0_8 | K_SETTLE1_8)) m |= C_FLAGS;
    return m;
}

// `prog`: 4-word instructions (word 0: opcode + settle bits [+ constant of FOLD_COL], 1 / 2: operands, 3: class mask), prog_len of them
// followed by END quadruples for the fetch-ahead.
template <bool FULL, bool ACC_MEM>
__global__ void __launch_bounds__(Q_THREADS) Q_OCC_ATTR
k_quotient_eval2(const uint32_t* __restrict__ prog, uint32_t prog_len, const Fr* const* __restrict__ cols, const QC29* __restrict__ consts,
                 const QC29* __restrict__ consts_rp /* the same constants in R' form */, const Fr* __restrict__ t_evals /* R' form */, uint32_t ext_k, uint32_t k,
                 Fr* __restrict__ out, Fr* tmp /* [slot][row] */, uint32_t* acc_mem /* ACC_MEM: [limb][row] */, uint32_t tile_alias /* measurement only, see the launch */,
                 const uint32_t* __restrict__ slice_tab /* per slice: first word of its instructions, their number */, uint32_t num_slices /* 0: one program for every workgroup */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // stack entry j < sp - 1 lives in LDS slot j ([slot][limb][lane]); entry sp - 1 is t0.  ONE address register (slot and lane; made opaque so
    // that the compiler does not split it into nine loop-invariant lane addresses + nine adds per access), the limb in the instruction's offset field.
    auto st_addr = [&](int slot) -> uint32_t {
        uint32_t a = (uint32_t)slot * (9u * Q_THREADS * 4u) + threadIdx.x * 4u;
        asm volatile("" : "+v"(a));
        return a;
    };
    auto st_put = [&](int slot, const Q29& v) {
        const uint32_t a = st_addr(slot);
#pragma unroll
        for (int q = 0; q < 9; ++q) *(uint32_t*)((char*)smem + a + q * (Q_THREADS * 4)) = v.l[q];
    };
    auto st_get = [&](int slot, Q29& v) {
        const uint32_t a = st_addr(slot);
#pragma unroll
        for (int q = 0; q < 9; ++q) v.l[q] = *(const uint32_t*)((const char*)smem + a + q * (Q_THREADS * 4));
    };
    const uint64_t ne = 1ull << ext_k;
    uint32_t blk = blockIdx.x;
    if (tile_alias) { const uint32_t xcd = blk & 7u, j = blk >> 3; blk = ((j >> tile_alias) << 3) | xcd; }       // 2^tile_alias consecutive workgroups of an XCD share one row tile (results wrong)
    if (num_slices) {
        // A sliced program (see zk_quotient_eval): workgroup ids go round the XCDs, so ids 8 j + x, j = t S ... t S + S - 1, are S workgroups that start
        // together on XCD x -- they take the S slices of ONE row tile and find each other's operands in that XCD's L2 / the Infinity Cache.
        const uint32_t xcd = blk & 7u, j = blk >> 3, sl = j % num_slices;
        blk = ((j / num_slices) << 3) | xcd;
        prog += slice_tab[2 * sl];
        prog_len = slice_tab[2 * sl + 1];
        out += (size_t)sl << ext_k;
    }
    const uint32_t i = blk * blockDim.x + threadIdx.x;
    const bool live = FULL || i < (uint32_t)ne;
    const uint32_t rot_scale = 1u << (ext_k - k), row_mask = (uint32_t)ne - 1u;
    Q29 t0, acc, B;
#pragma unroll
    for (int q = 0; q < 9; ++q) { t0.l[q] = 0; acc.l[q] = 0; B.l[q] = 0; }        // acc: registers only when !ACC_MEM
    int sp = 0;
    typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
    using GU4 = const U32x4 __attribute__((address_space(1)));
    using GU1 = uint32_t __attribute__((address_space(1)));
    // The raw memory operand stays in the two 4-register tuples the loads write (unpacked from there: no copies between the load and its use)
    struct Raw { U32x4 lo, hi; };
    auto load = [&](uint32_t a, uint32_t b, Raw& r) {
        const uint32_t row = (i + b * rot_scale) & row_mask;          // b = the rotation as a two's complement word: wraps like the domain
        if (FULL || live) {
            GU4* q = (GU4*)(uintptr_t)(cols[a] + row);
            r.lo = q[0]; r.hi = q[1];
        }
    };
    auto raw_fr = [](Raw& r) -> Fr {        // opaque: each unpack site unpacks for itself (the compiler otherwise unpacks on EVERY instruction and copies)
        asm volatile("" : "+v"(r.lo), "+v"(r.hi));
        Fr x;
        x.l[0] = r.lo.x; x.l[1] = r.lo.y; x.l[2] = r.lo.z; x.l[3] = r.lo.w;
        x.l[4] = r.hi.x; x.l[5] = r.hi.y; x.l[6] = r.hi.z; x.l[7] = r.hi.w;
        return x;
    };
    auto cst = [&](const QC29* __restrict__ tab, uint32_t j) -> Q29 {       // wave-uniform index: scalar loads, the limbs stay in scalar registers
        Q29 r;
#pragma unroll
        for (int q = 0; q < 9; ++q) r.l[q] = tab[j].l[q];
        return r;
    };
    // acc <- acc * c + v  (v: limbs below 3 x 2^29)
    auto fold = [&](uint32_t w0_, const Q29& c, const Q29& v) {
        if (!ACC_MEM) {
            mul29_ub_ipa(acc, c);
#pragma unroll
            for (int q = 0; q < 9; ++q) acc.l[q] += v.l[q];
            return;
        }
        Q29 a;
#pragma unroll
        for (int q = 0; q < 9; ++q) a.l[q] = 0;
        uint32_t ii = i;
        asm volatile("" : "+v"(ii));              // the nine limb addresses are made here, not kept in eighteen registers across the loop
        GU1* qa = (GU1*)(uintptr_t)(acc_mem + ii);
        if (!(w0_ & K_FIRST_FOLD) && live) {
#pragma unroll
            for (int l = 0; l < 9; ++l) a.l[l] = qa[(uint64_t)l << ext_k];
        }
        mul29_ub_ipa(a, c);
        if (live) {
#pragma unroll
            for (int l = 0; l < 9; ++l) qa[(uint64_t)l << ext_k] = a.l[l] + v.l[l];
        }
    };
    uint32_t w0 = prog[0], w1 = prog[1], w2 = prog[2], w3 = prog[3];
    uint32_t n0 = prog[4], n1 = prog[5], n2 = prog[6], n3 = prog[7];
    Raw m;
    m.lo = U32x4{0, 0, 0, 0}; m.hi = U32x4{0, 0, 0, 0};
    if (w3 & C_HASMEM) load(w1, w2, m);
    for (uint32_t pc = 0; pc < prog_len; ++pc) {
        const uint32_t f0 = prog[4 * pc + 8], f1 = prog[4 * pc + 9], f2 = prog[4 * pc + 10], f3 = prog[4 * pc + 11];
        if ((w0 & 0xffu) == Q_END) break;
        // 1. operands: a memory operand leaves its raw registers (which then take the load for instruction pc + 1), a binary stack operation
        //    takes the entry below the top
        if (w3 & C_SPILL) { if (sp >= 1) st_put(sp - 1, t0); ++sp; }
        if (w3 & C_UNPACK_T) q_unpack_to(t0, raw_fr(m));
        if (w3 & C_UNPACK_B) q_unpack_to(B, raw_fr(m));
        if (w3 & C_UNPACK_B32) q_unpack_x32_to(B, raw_fr(m));
        if (w3 & C_POP_B) { st_get(sp - 2, B); --sp; }
        if (n3 & C_HASMEM) load(n1, n2, m);
        // 2. settle / carry-propagation requests of the lowering (bit 0: the top, bit 1: the entry below it = B)
        if (w3 & C_FLAGS) {
            if (w0 & K_SETTLE0) q_settle_ip(t0);
            if (w0 & K_SETTLE1) q_settle_ip(B);
            if (w0 & K_NORM0) normalize29(t0);
            if (w0 & K_NORM1) normalize29(B);
            if (w0 & K_SETTLE0_8) q_settle8_ip(t0);
            if (w0 & K_SETTLE1_8) q_settle8_ip(B);
        }
        // 3. the operation, on t0 in place
        if (w3 & C_MULV) mul29_ipa(t0, B);
        if (w3 & C_MULC) mul29_ub_ipa(t0, cst(consts_rp, w1));
        if (w3 & C_ADDV) {
#pragma unroll
            for (int q = 0; q < 9; ++q) t0.l[q] += B.l[q];
        }
        if (w3 & C_SUBV) {                                                  // canonical subtrahend: its top limb is below that of 2p, no borrow
#pragma unroll
            for (int q = 0; q < 9; ++q) t0.l[q] = t0.l[q] + kp_balanced<Fr29P>(2, q) - B.l[q];
        }
        if (w3 & C_FOLDC) fold(w0, cst(consts_rp, w0 >> K_CONST_SHIFT), B);
        if (w3 & C_RARE) {
            if (w3 & C_PUSHC) {
                const Q29 c = cst(consts, w1);
#pragma unroll
                for (int q = 0; q < 9; ++q) t0.l[q] = c.l[q];
            }
            if (w3 & C_ADDC) {
                const Q29 c = cst(consts, w1);
#pragma unroll
                for (int q = 0; q < 9; ++q) t0.l[q] += c.l[q];
            }
            if (w3 & C_RSUB) {                                              // B - t0 (t0 settled): carry the borrow of the top limb out before anyone multiplies
#pragma unroll
                for (int q = 0; q < 9; ++q) t0.l[q] = B.l[q] + kp_balanced<Fr29P>(2, q) - t0.l[q];
                normalize29(t0);
            }
            if (w3 & C_MULS) { q_shl5_ip(t0); mul29_ipb(t0, B); }            // t0 <- mul29(B, 32 t0)
            if (w3 & C_NEG) {
#pragma unroll
                for (int q = 0; q < 9; ++q) t0.l[q] = kp_balanced<Fr29P>(2, q) - t0.l[q];
                normalize29(t0);
            }
            if (w3 & C_SQ) { Q29 s2 = t0; q_shl5_ip(s2); mul29_ipa(t0, s2); }
            if (w3 & C_DBL) {
#pragma unroll
                for (int q = 0; q < 9; ++q) t0.l[q] <<= 1;
            }
            if (w3 & C_TEE) { if (live) stg(tmp + (((uint64_t)w1 << ext_k) + i), pack29_lt2p(t0)); }
            if (w3 & C_FOLD) { fold(w0, cst(consts_rp, w1), t0); --sp; if (sp >= 1) st_get(sp - 1, t0); }
        }
        w0 = n0; w1 = n1; w2 = n2; w3 = n3;
        n0 = f0; n1 = f1; n2 = f2; n3 = f3;
    }
    if (live) {
        Q29 a = acc;
        if (ACC_MEM) {
            GU1* qa = (GU1*)(uintptr_t)(acc_mem + i);
#pragma unroll
            for (int l = 0; l < 9; ++l) a.l[l] = qa[(uint64_t)l << ext_k];
        }
        if (t_evals) stg(out + i, pack29_lt2p(mul29(a, unpack29<Fr29P>(ldg(t_evals + (i & (rot_scale - 1u)))))));
        else { normalize29(a); stg(out + i, reduce_lazy29(a)); }
    }
}

// ---- lowering: caller's postfix program -> the kernel's instruction stream -------------------------------------
struct LNode {
    enum Kind : uint8_t { MEM, CONST, UN, BIN, CONSTOP, TEE, MAT } kind;
    bool has_tee;           // the subtree parks an intermediate (evaluation order then matters for readers of that slot)
    bool is_tmp;            // MEM: a parked intermediate read back
    uint32_t op, a, b;      // UN / BIN / CONSTOP: opcode; MEM: column, rotation; CONST / CONSTOP: constant; TEE: slot
    int32_t x, y;
};
struct LowInstr { uint32_t w0, a, b; };

// expression trees of the program, re-emitted with memory operands (see the header comment)
static void lower_fuse(const uint32_t* prog, uint32_t len, uint32_t num_cols, std::vector<LowInstr>* out) {
    std::vector<LNode> nodes;
    nodes.reserve(len);
    std::vector<int32_t> st;
    auto add = [&](LNode n) { nodes.push_back(n); return (int32_t)nodes.size() - 1; };
    auto leafish = [&](int32_t n) { return nodes[n].kind == LNode::MEM || nodes[n].kind == LNode::CONST; };
    struct Work { int32_t node; int plan; };
    std::vector<Work> work;
    // plans of a BIN node after its operands: 0 = stack op, 1 = y from memory, 2 = y constant, 3 = x from memory (operands swapped), 4 = x constant (swapped)
    auto emit = [&](int32_t root) {
        work.push_back({root, -1});
        while (!work.empty()) {
            const Work wk = work.back();
            work.pop_back();
            const LNode& n = nodes[wk.node];
            switch (n.kind) {
                case LNode::MAT: break;
                case LNode::MEM: out->push_back({Q_PUSH_COL, n.a, n.b}); break;
                case LNode::CONST: out->push_back({Q_PUSH_CONST, n.a, 0}); break;
                case LNode::UN:
                    if (wk.plan < 0) { work.push_back({wk.node, 0}); work.push_back({n.x, -1}); }
                    else out->push_back({n.op, 0, 0});
                    break;
                case LNode::CONSTOP:
                    if (wk.plan < 0) { work.push_back({wk.node, 0}); work.push_back({n.x, -1}); }
                    else out->push_back({n.op, n.a, 0});
                    break;
                case LNode::TEE:
                    if (wk.plan < 0) { work.push_back({wk.node, 0}); work.push_back({n.x, -1}); }
                    else out->push_back({Q_TEE_TMP, n.a, 0});
                    break;
                case LNode::BIN: {
                    const LNode &x = nodes[n.x], &y = nodes[n.y];
                    if (wk.plan < 0) {
                        int plan = 0;
                        if (y.kind == LNode::MEM) plan = 1;
                        else if (y.kind == LNode::CONST && (n.op == Q_ADD || n.op == Q_MUL)) plan = 2;
                        else if (x.kind == LNode::MEM && !leafish(n.y) && !(x.is_tmp && y.has_tee)) plan = 3;
                        else if (x.kind == LNode::CONST && !leafish(n.y) && (n.op == Q_ADD || n.op == Q_MUL)) plan = 4;
                        work.push_back({wk.node, plan});
                        if (plan == 0) { work.push_back({n.y, -1}); work.push_back({n.x, -1}); }
                        else if (plan == 1 || plan == 2) work.push_back({n.x, -1});
                        else work.push_back({n.y, -1});
                    } else if (wk.plan == 0) out->push_back({n.op, 0, 0});
                    else if (wk.plan == 1) out->push_back({n.op == Q_ADD ? K_ADD_COL : n.op == Q_SUB ? K_SUB_COL : K_MUL_COL, y.a, y.b});
                    else if (wk.plan == 2) out->push_back({n.op == Q_ADD ? Q_ADD_CONST : Q_MUL_CONST, y.a, 0});
                    else if (wk.plan == 3) out->push_back({n.op == Q_ADD ? K_ADD_COL : n.op == Q_SUB ? K_RSUB_COL : K_MUL_COL, x.a, x.b});
                    else out->push_back({n.op == Q_ADD ? Q_ADD_CONST : Q_MUL_CONST, x.a, 0});
                    break;
                }
            }
        }
    };
    auto materialise_pending = [&]() {
        for (int32_t& e : st)
            if (nodes[e].kind != LNode::MAT) { emit(e); e = add({LNode::MAT, false, false, 0, 0, 0, -1, -1}); }
    };
    for (uint32_t pc = 0; pc < len; ++pc) {
        const uint32_t op = prog[3 * pc], a = prog[3 * pc + 1], b = prog[3 * pc + 2];
        if (op == Q_END) break;
        switch (op) {
            case Q_PUSH_COL: st.push_back(add({LNode::MEM, false, false, 0, a, b, -1, -1})); break;
            case Q_PUSH_TMP: st.push_back(add({LNode::MEM, false, true, 0, num_cols + a, 0, -1, -1})); break;     // parked intermediates are columns num_cols + slot
            case Q_PUSH_CONST: st.push_back(add({LNode::CONST, false, false, 0, a, 0, -1, -1})); break;
            case Q_ADD: case Q_SUB: case Q_MUL: {
                const int32_t y = st.back(); st.pop_back();
                const int32_t x = st.back(); st.pop_back();
                st.push_back(add({LNode::BIN, nodes[x].has_tee || nodes[y].has_tee, false, op, 0, 0, x, y}));
                break;
            }
            case Q_NEG: case Q_SQUARE: case Q_DOUBLE: { const int32_t x = st.back(); st.back() = add({LNode::UN, nodes[x].has_tee, false, op, 0, 0, x, -1}); break; }
            case Q_MUL_CONST: case Q_ADD_CONST: { const int32_t x = st.back(); st.back() = add({LNode::CONSTOP, nodes[x].has_tee, false, op, a, 0, x, -1}); break; }
            case Q_TEE_TMP: { const int32_t x = st.back(); st.back() = add({LNode::TEE, true, false, 0, a, 0, x, -1}); break; }
            case Q_FOLD: {
                const int32_t r = st.back(); st.pop_back();
                materialise_pending();
                if (nodes[r].kind == LNode::MEM && a < (1u << (32 - K_CONST_SHIFT))) out->push_back({K_FOLD_COL | (a << K_CONST_SHIFT), nodes[r].a, nodes[r].b});
                else { emit(r); out->push_back({Q_FOLD, a, 0}); }
                break;
            }
            default: break;     // validate_program has refused everything else
        }
    }
    materialise_pending();
}

// Settle / normalise bits, prefetch hazards and stack depth of a fused program (bounds: header comment).  V in units of p, L in
// units of 2^29.  Round 6: a stack entry may grow to (8, 4) -- sums only ever need limbs below 2^31 (carries are propagated where a
// limb bound is in the way: K_NORM, 24 instructions) and the consumers that care about the VALUE say so: the first operand of a
// product by a memory operand takes V <= 5, by a constant anything here, the second operand of a stack product / a subtrahend / a
// value being parked must be settled.  A value above 4p that has to be settled after all takes q_settle8 (one more conditional
// subtraction).  The Horner steps S = S y^g + term of the compiled class programs run without a single settle that way; round 5's
// rule (everything within (4, 4), settle on every excess: ZK_QUOTIENT_RELAXED=0) spent 8 024 settles on the 12 289 products of the
// EVM-style class program.
static int lower_bounds(std::vector<LowInstr>* prog_io, uint32_t num_cols, int* max_depth) {
    struct Bd { int V, L; };
    std::vector<Bd> bs;
    std::vector<LowInstr> out;
    out.reserve(prog_io->size() + 8);
    int mx = 0;
    const char* renv = getenv("ZK_QUOTIENT_RELAXED");
    const bool relaxed = !(renv && atoi(renv) == 0);
    const int capV = relaxed ? 8 : 4;
    for (size_t ii = 0; ii < prog_io->size(); ++ii) {
        LowInstr in = (*prog_io)[ii];
        const uint32_t op = in.w0 & 0xffu;
        auto settle0 = [&]() { in.w0 &= ~(K_NORM0 | K_SETTLE0 | K_SETTLE0_8); in.w0 |= bs.back().V > 4 ? K_SETTLE0_8 : K_SETTLE0; bs.back() = {2, 1}; };
        auto settle1 = [&]() { in.w0 &= ~(K_NORM1 | K_SETTLE1 | K_SETTLE1_8); in.w0 |= bs[bs.size() - 2].V > 4 ? K_SETTLE1_8 : K_SETTLE1; bs[bs.size() - 2] = {2, 1}; };
        auto norm0 = [&]() { if (!relaxed) { settle0(); return; } if (!(in.w0 & (K_SETTLE0 | K_SETTLE0_8))) in.w0 |= K_NORM0; bs.back().L = 1; };
        auto norm1 = [&]() { if (!relaxed) { settle1(); return; } if (!(in.w0 & (K_SETTLE1 | K_SETTLE1_8))) in.w0 |= K_NORM1; bs[bs.size() - 2].L = 1; };
        auto settled0 = [&]() { if (bs.back().V > 2 || bs.back().L > 1) settle0(); };          // t0 must be a settled value
        // t0 op= something of bounds (dV, dL): make room in t0
        auto room0 = [&](int dV, int dL) {
            if (bs.back().L + dL > 4) norm0();
            if (bs.back().V + dV > capV) settle0();
        };
        const size_t need = (op == Q_ADD || op == Q_SUB || op == Q_MUL) ? 2 : (op == Q_PUSH_COL || op == Q_PUSH_CONST || op == K_FOLD_COL || op == K_NOP) ? 0 : 1;
        if (bs.size() < need) return -1;
        switch (op) {
            case Q_PUSH_COL: case Q_PUSH_CONST: bs.push_back({1, 1}); break;
            case Q_ADD: {
                Bd &x = bs[bs.size() - 2], &y = bs.back();
                if (x.L + y.L > 4) { if (y.L >= x.L) norm0(); else norm1(); }
                if (x.L + y.L > 4) { if (y.L > 1) norm0(); else norm1(); }
                if (x.V + y.V > capV) { if (y.V >= x.V) settle0(); else settle1(); }
                if (x.V + y.V > capV) { if (y.V > 2) settle0(); else settle1(); }
                if (x.V + y.V > capV || x.L + y.L > 4) return -1;
                const Bd r{x.V + y.V, x.L + y.L};
                bs.pop_back(); bs.back() = r;
                break;
            }
            case Q_SUB: {
                settled0();
                if (bs[bs.size() - 2].L + 2 > 4) norm1();
                if (bs[bs.size() - 2].V + 2 > capV) settle1();
                const Bd r{bs[bs.size() - 2].V + 2, 1};
                bs.pop_back(); bs.back() = r;
                break;
            }
            case Q_MUL: {
                settled0();
                if (bs[bs.size() - 2].V * bs.back().V > 5) settle1();
                bs.pop_back(); bs.back() = {2, 1};
                break;
            }
            case Q_NEG: settled0(); bs.back() = {3, 1}; break;       // 2p - t0 reaches 2p itself (t0 = 0): not below 2p
            case Q_SQUARE: settled0(); bs.back() = {2, 1}; break;
            case Q_DOUBLE: if (bs.back().L > 2) norm0(); if (2 * bs.back().V > capV) settle0(); bs.back() = {2 * bs.back().V, 2 * bs.back().L}; break;
            case Q_FOLD: if (bs.back().L > 3) norm0(); bs.pop_back(); break;
            case Q_MUL_CONST: bs.back() = {2, 1}; break;
            case Q_ADD_CONST: case K_ADD_COL: room0(1, 1); bs.back() = {bs.back().V + 1, bs.back().L + 1}; break;
            case Q_TEE_TMP: settled0(); break;
            case K_SUB_COL: room0(2, 2); bs.back() = {bs.back().V + 2, bs.back().L + 2}; break;
            case K_RSUB_COL: settled0(); bs.back() = {3, 1}; break;
            case K_MUL_COL: if (bs.back().V > 5) settle0(); bs.back() = {2, 1}; break;
            case K_FOLD_COL: case K_NOP: break;
            default: return -1;
        }
        // the memory operand of an instruction is loaded while its predecessor runs: a parked intermediate must not be
        // read back by the instruction right behind the one that parks it
        if (k_has_mem_host(in.w0) && in.a >= num_cols) {
            for (int back = 1; back <= Q_PREFETCH_DIST; ++back) {
                if ((int)out.size() < back) break;
                const LowInstr& pv = out[out.size() - back];
                if ((pv.w0 & 0xffu) == Q_TEE_TMP && pv.a == in.a - num_cols) { for (int q = back; q <= Q_PREFETCH_DIST; ++q) out.push_back({K_NOP, 0, 0}); break; }
            }
        }
        out.push_back(in);
        if ((int)bs.size() > mx) mx = (int)bs.size();
    }
    *max_depth = mx;
    prog_io->swap(out);
    return 0;
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

// The lowering as a host-only entry point (no device involved): what the kernel will run for a program.  Used by the
// CPU tests, which execute the lowered stream limb by limb and check every bound the kernel relies on.
extern "C" int zk_host_quotient_lower(const uint32_t* h_program, uint32_t num_instr, uint32_t num_cols, int fuse, uint32_t* out_words, size_t cap_words,
                                      uint32_t* out_instr, int* out_depth) {
    if (!h_program || !out_instr || !out_depth) return ZK_ERR_INVALID_ARG;
    {   // stack discipline of the caller's program (zk_quotient_eval checks the same, with messages, in validate_program)
        int sp = 0;
        for (uint32_t pc = 0; pc < num_instr && h_program[3 * pc] != Q_END; ++pc) {
            const uint32_t op = h_program[3 * pc];
            if (op == Q_PUSH_COL || op == Q_PUSH_CONST || op == Q_PUSH_TMP) ++sp;
            else if (op == Q_ADD || op == Q_SUB || op == Q_MUL) { if (sp < 2) return ZK_ERR_INVALID_ARG; --sp; }
            else if (op == Q_FOLD) { if (sp < 1) return ZK_ERR_INVALID_ARG; --sp; }
            else if (op == Q_NEG || op == Q_SQUARE || op == Q_DOUBLE || op == Q_MUL_CONST || op == Q_ADD_CONST || op == Q_TEE_TMP) { if (sp < 1) return ZK_ERR_INVALID_ARG; }
            else return ZK_ERR_INVALID_ARG;
        }
    }
    std::vector<LowInstr> low;
    if (fuse) lower_fuse(h_program, num_instr, num_cols, &low);
    else
        for (uint32_t pc = 0; pc < num_instr && h_program[3 * pc] != Q_END; ++pc) {
            const uint32_t op = h_program[3 * pc], a_ = h_program[3 * pc + 1], b_ = h_program[3 * pc + 2];
            if (op == Q_PUSH_TMP) low.push_back({Q_PUSH_COL, num_cols + a_, 0});
            else low.push_back({op, a_, b_});
        }
    int depth = 0;
    if (lower_bounds(&low, num_cols, &depth)) return ZK_ERR_INVALID_ARG;
    *out_instr = (uint32_t)low.size();
    *out_depth = depth;
    if (out_words) {
        if (cap_words < low.size() * 3) return ZK_ERR_INVALID_ARG;
        for (size_t i = 0; i < low.size(); ++i) { out_words[3 * i] = low[i].w0; out_words[3 * i + 1] = low[i].a; out_words[3 * i + 2] = low[i].b; }
    }
    return ZK_OK;
}

// ---- slicing a large program for the caches (round 6) --------------------------------------------------------------------------
// A class program of the EVM-style constraint system reads each of its operands ~70 times per row, hundreds of instructions apart, and the
// rows of the ~5 000 resident waves x their few hundred columns are gigabytes: every read goes to HBM (5 TB/s of operand loads for 5 KB of
// distinct operands per row).  Made to evaluate ONE row tile with four to sixteen workgroups at a time (ZK_QUOTIENT_TILE_ALIAS) the same
// launch takes 116 ... 107 ms instead of 139: with a sixteenth of the rows in flight their operands stay in the Infinity Cache.  So a program
// that is a top-level sum  acc = acc * c_f + term_f  is cut at fold boundaries into S slices, slice s is evaluated from acc = 0 by its own
// workgroup over the same rows at the same time (k_quotient_eval2: slice_tab), and
//      acc_out(slice s) = acc_in * prod_{f in s} c_f + P_s
// puts the partial sums P_s together afterwards (an S-term FOLD_COL chain through this same function).  A cut is only made where no parked
// intermediate is alive; parking slots are renumbered per slice (slices run concurrently).  ZK_QUOTIENT_SLICES=0 turns it off, =N asks for N.
struct SlicePlan { std::vector<uint32_t> cut; };       // cut[s] .. cut[s + 1]: the caller's instructions of slice s
static bool plan_slices(const uint32_t* prog, uint32_t len, uint32_t ext_k, SlicePlan* out) {
    const int knob = getenv("ZK_QUOTIENT_SLICES") ? atoi(getenv("ZK_QUOTIENT_SLICES")) : -1;        // read per call: the tests flip it
    if (knob == 0 || (knob < 0 && (len < 2048 || ext_k < 16)) || ext_k < 11) return false;     // a forced count (tests) slices small programs too
    std::vector<uint32_t> bpos;           // instruction index behind a top-level FOLD
    std::vector<uint64_t> bcost;          // cost of everything in front of it
    std::vector<int32_t> blocked(len + 2, 0);
    std::vector<uint32_t> last_tee;
    std::vector<uint64_t> ops;
    std::vector<uint32_t> colset;
    uint64_t cost = 0, reads = 0;
    int sp = 0;
    uint32_t n = 0;
    for (; n < len && prog[3 * n] != Q_END; ++n) {
        const uint32_t op = prog[3 * n], a = prog[3 * n + 1];
        switch (op) {
            case Q_PUSH_COL: ++sp; cost += 3; ++reads; ops.push_back(((uint64_t)a << 32) | prog[3 * n + 2]); colset.push_back(a); break;
            case Q_PUSH_CONST: ++sp; cost += 1; break;
            case Q_PUSH_TMP:
                ++sp; cost += 3;
                if (a < last_tee.size()) { ++blocked[last_tee[a] + 1]; --blocked[n + 1]; }       // no cut between the parking and this read
                break;
            case Q_TEE_TMP: if (a >= last_tee.size()) last_tee.resize(a + 1, 0); last_tee[a] = n; cost += 4; break;
            case Q_ADD: case Q_SUB: --sp; cost += 1; break;
            case Q_MUL: --sp; cost += 24; break;
            case Q_SQUARE: case Q_MUL_CONST: cost += 24; break;
            case Q_FOLD: --sp; cost += 24; if (sp == 0) { bpos.push_back(n + 1); bcost.push_back(cost); } break;
            default: cost += 1; break;
        }
    }
    if (bpos.size() < 2 || bpos.back() != n) return false;          // not a sum of terms (or something trails the last fold)
    std::sort(ops.begin(), ops.end()); ops.erase(std::unique(ops.begin(), ops.end()), ops.end());
    std::sort(colset.begin(), colset.end()); colset.erase(std::unique(colset.begin(), colset.end()), colset.end());
    if (knob < 0 && reads < 4 * ops.size()) return false;           // operands read once or twice: nothing to keep in a cache
    // slices wanted: the operands of the rows in flight (~5 000 waves x 64) should be well inside the 256 MiB Infinity Cache
    const uint64_t rows_in_flight = std::min<uint64_t>(1ull << ext_k, 327680);
    const uint64_t ws = colset.size() * 32ull * rows_in_flight;
    uint32_t want = knob > 0 ? (uint32_t)knob : (uint32_t)std::min<uint64_t>(64, (ws + (96ull << 20) - 1) / (96ull << 20));
    if (want < 2) return false;
    // cuts at unblocked boundaries, by cost
    std::vector<char> ok(bpos.size(), 1);
    {
        int32_t run = 0;
        size_t b = 0;
        for (uint32_t q = 0; q <= n && b < bpos.size(); ++q) {
            run += blocked[q];
            if (bpos[b] == q) { ok[b] = run == 0; ++b; }
        }
    }
    out->cut.assign(1, 0);
    uint32_t made = 1;
    for (size_t b = 0; b + 1 < bpos.size() && made < want; ++b) {
        if (!ok[b]) continue;
        if (bcost[b] * want >= cost * made) { out->cut.push_back(bpos[b]); ++made; }
    }
    out->cut.push_back(n);
    return out->cut.size() > 2;
}

extern "C" int zk_host_quotient_slices(const uint32_t* program, uint32_t num_instr, uint32_t ext_k, uint32_t* out_cuts, size_t cap, uint32_t* out_count) {
    if (!program || !out_count) return ZK_ERR_INVALID_ARG;
    SlicePlan plan;
    *out_count = 0;
    if (!plan_slices(program, num_instr, ext_k, &plan)) return ZK_OK;
    *out_count = (uint32_t)plan.cut.size();
    if (out_cuts) {
        if (cap < plan.cut.size()) return ZK_ERR_INVALID_ARG;
        for (size_t i = 0; i < plan.cut.size(); ++i) out_cuts[i] = plan.cut[i];
    }
    return ZK_OK;
}

extern "C" int zk_quotient_eval(zk_ctx* ctx, const uint32_t* h_program, uint32_t num_instr, const void* const* h_col_ptrs, uint32_t num_cols,
                                const void* h_consts, uint32_t num_consts, uint32_t k, uint32_t ext_k, int divide_by_vanishing, void* d_out) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, h_program && d_out && (h_col_ptrs || !num_cols) && (h_consts || !num_consts), "null pointer");
    ZK_REQUIRE(ctx, k <= ext_k && ext_k <= 28, "need k <= ext_k <= 28");
    int depth = 0;
    uint32_t num_tmp = 0;
    int rc = validate_program(ctx, h_program, num_instr, num_cols, num_consts, &depth, &num_tmp);
    if (rc) return rc;
    const int kernel_knob = getenv("ZK_QUOTIENT_KERNEL") ? atoi(getenv("ZK_QUOTIENT_KERNEL")) : 2;       // 2: fixed register roles (k_quotient_eval2: one stack entry in registers, 4-word instructions); 1: round 5's kernel
    const bool v2 = kernel_knob != 1;
    const uint64_t ne = 1ull << ext_k;
    // a large sum of terms whose operands are read many times is cut into slices that share their rows' operands through the caches (plan_slices)
    SlicePlan plan;
    const bool sliced = v2 && ((ne / Q_THREADS) & 7) == 0 && plan_slices(h_program, num_instr, ext_k, &plan);
    const uint32_t S = sliced ? (uint32_t)plan.cut.size() - 1 : 1;
    // lower the program for the kernel (memory operands, settle bits, prefetch hazards); ZK_QUOTIENT_FUSE=0 keeps the
    // caller's instruction sequence (measurement knob: only the bounds pass runs)
    std::vector<std::vector<LowInstr>> lows(S);
    std::vector<Fr> slice_k;                  // sliced: the product of a slice's fold constants (what the accumulator coming in is multiplied by)
    if (sliced) {
        uint32_t next_slot = 0;
        int dmax = 0;
        slice_k.assign(S, Fr::one());
        for (uint32_t sl = 0; sl < S; ++sl) {
            std::vector<uint32_t> sub(h_program + 3 * plan.cut[sl], h_program + 3 * plan.cut[sl + 1]);
            std::unordered_map<uint32_t, uint32_t> slot_of;        // parking slots of this slice -> its own range (slices run at the same time)
            int spd = 0;
            for (size_t q = 0; q < sub.size() / 3; ++q) {
                const uint32_t op = sub[3 * q];
                if (op == Q_TEE_TMP || op == Q_PUSH_TMP) {
                    auto it = slot_of.find(sub[3 * q + 1]);
                    if (it == slot_of.end()) {
                        if (op == Q_PUSH_TMP) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient program: slice %u reads an intermediate parked outside it", sl);
                        it = slot_of.emplace(sub[3 * q + 1], next_slot++).first;
                    }
                    sub[3 * q + 1] = it->second;
                }
                if (op == Q_PUSH_COL || op == Q_PUSH_CONST || op == Q_PUSH_TMP) ++spd;
                else if (op == Q_ADD || op == Q_SUB || op == Q_MUL) --spd;
                else if (op == Q_FOLD) { --spd; if (spd == 0) slice_k[sl] = slice_k[sl] * ((const Fr*)h_consts)[sub[3 * q + 1]]; }
            }
            sub.push_back(Q_END); sub.push_back(0); sub.push_back(0);
            lower_fuse(sub.data(), (uint32_t)(sub.size() / 3), num_cols, &lows[sl]);
            int dsl = 0;
            if (lower_bounds(&lows[sl], num_cols, &dsl)) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient program: lowering failed");
            dmax = std::max(dmax, dsl);
        }
        depth = dmax;
        num_tmp = next_slot;
    } else {
        std::vector<LowInstr>& low = lows[0];
        const char* env = getenv("ZK_QUOTIENT_FUSE");
        if (env && atoi(env) == 0) {
            for (uint32_t pc = 0; pc < num_instr && h_program[3 * pc] != Q_END; ++pc) {
                const uint32_t op = h_program[3 * pc], a_ = h_program[3 * pc + 1], b_ = h_program[3 * pc + 2];
                if (op == Q_PUSH_TMP) low.push_back({Q_PUSH_COL, num_cols + a_, 0});
                else low.push_back({op, a_, b_});
            }
        } else lower_fuse(h_program, num_instr, num_cols, &low);
        if (lower_bounds(&low, num_cols, &depth)) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient program: lowering failed");
    }
    if (depth > Q_MAX_STACK) return ctx->fail(ZK_ERR_UNSUPPORTED, "quotient program: stack deeper than %d", Q_MAX_STACK);
    if (depth < 1) depth = 1;
    Fr* d_tmp = nullptr;
    if (num_tmp) {
        d_tmp = (Fr*)ctx->get_scratch(SC_QTMP, ((size_t)num_tmp << ext_k) * sizeof(Fr));
        if (!d_tmp) return ZK_ERR_OOM;
    }
    uint32_t low_len = 0;
    // the accumulator's home: registers for programs that fold on most instructions (and for slices), memory for the rest (see the kernel)
    uint32_t folds = 0;
    for (std::vector<LowInstr>& low : lows) {
        bool first = true;
        for (LowInstr& in : low) {
            const uint32_t o = in.w0 & 0xffu;
            if (o == Q_FOLD || o == K_FOLD_COL) { if (first) in.w0 |= K_FIRST_FOLD; first = false; ++folds; }
        }
        low_len += (uint32_t)low.size();
    }
    static const int acc_knob = getenv("ZK_QUOTIENT_ACC_MEM") ? atoi(getenv("ZK_QUOTIENT_ACC_MEM")) : -1;      // measurement knob: 0 / 1 force the variant
    const bool acc_in_mem = !sliced && folds > 0 && (acc_knob < 0 ? (uint64_t)folds * 16 <= low_len : acc_knob == 1);
    uint32_t* d_acc = nullptr;
    if (acc_in_mem) {
        d_acc = (uint32_t*)ctx->get_scratch(SC_QACC, ((size_t)9 << ext_k) * sizeof(uint32_t));
        if (!d_acc) return ZK_ERR_OOM;
    }
    Fr* d_part = nullptr;                     // sliced: the slices' sums, [slice][row]
    if (sliced) {
        d_part = (Fr*)ctx->get_scratch(SC_QSLICE, ((size_t)S << ext_k) * sizeof(Fr));
        if (!d_part) return ZK_ERR_OOM;
    }
    if (getenv("ZK_QUOTIENT_TRACE") && low_len >= 64) {        // what the kernel will run: lowered instructions by opcode, settle bits, stack depth
        static const char* names[22] = {"END", "PUSH_COL", "PUSH_CONST", "ADD", "SUB", "MUL", "NEG", "SQUARE", "DOUBLE", "FOLD", "MUL_CONST", "ADD_CONST", "TEE_TMP", "PUSH_TMP", "?", "?",
                                        "ADD_COL", "SUB_COL", "RSUB_COL", "MUL_COL", "FOLD_COL", "NOP"};
        uint32_t hist[22] = {0}, settles = 0, norms = 0;
        for (const std::vector<LowInstr>& low : lows) for (const LowInstr& in : low) { const uint32_t o = in.w0 & 0xffu; if (o < 22) ++hist[o]; settles += ((in.w0 & (K_SETTLE0 | K_SETTLE0_8)) ? 1 : 0) + ((in.w0 & (K_SETTLE1 | K_SETTLE1_8)) ? 1 : 0); norms += ((in.w0 & K_NORM0) ? 1 : 0) + ((in.w0 & K_NORM1) ? 1 : 0); }
        fprintf(stderr, "[zk quotient] 2^%u rows, %u lowered instructions%s, depth %d, %u settles, %u carry propagations:", ext_k, low_len, sliced ? (" in " + std::to_string(S) + " slices").c_str() : "", depth, settles, norms);
        for (int o = 0; o < 22; ++o) if (hist[o]) fprintf(stderr, " %s %u", names[o], hist[o]);
        if (sliced) { fprintf(stderr, "; slice lengths"); for (const std::vector<LowInstr>& low : lows) fprintf(stderr, " %zu", low.size()); fprintf(stderr, "; %u parking slots", num_tmp); }
        fprintf(stderr, "\n");
    }
    std::vector<Fr> tev;
    if (divide_by_vanishing) vanishing_inverses(k, ext_k, &tev);
    // constants that multiply (MUL_CONST, FOLD) and the vanishing inverses go to the device in R' = 2^261 form too: x 32
    auto times32 = [](Fr x) { for (int j = 0; j < 5; ++j) x = dbl(x); return x; };
    std::vector<Fr> consts_r((const Fr*)h_consts, (const Fr*)h_consts + num_consts);
    if (sliced) consts_r.insert(consts_r.end(), slice_k.begin(), slice_k.end());      // constants num_consts + s: what the accumulator coming into slice s is multiplied by
    const uint32_t nc_all = (uint32_t)consts_r.size();
    std::vector<Fr> consts_rp(consts_r);
    for (Fr& c : consts_rp) c = times32(c);
    for (Fr& t : tev) t = times32(t);
    // column table = the caller's columns, then one pseudo-column per parked intermediate
    std::vector<const void*> col_tab(h_col_ptrs, h_col_ptrs + num_cols);
    for (uint32_t t = 0; t < num_tmp; ++t) col_tab.push_back(d_tmp + ((size_t)t << ext_k));
    // sliced: the partial sums are columns num_cols + num_tmp + s of a last little program, acc = acc * k_s + P_s over the slices (and the vanishing division the
    // caller asked for), which rides in the same upload and is launched right behind the slices: no second call, no host synchronisation in between
    std::vector<LowInstr> comb;
    if (sliced) {
        if (nc_all >= (1u << (32 - K_CONST_SHIFT))) return ctx->fail(ZK_ERR_UNSUPPORTED, "quotient program: too many constants for a sliced launch");
        for (uint32_t sl = 0; sl < S; ++sl) {
            col_tab.push_back(d_part + ((size_t)sl << ext_k));
            comb.push_back({K_FOLD_COL | ((num_consts + sl) << K_CONST_SHIFT) | (sl ? 0u : K_FIRST_FOLD), num_cols + num_tmp + sl, 0});
        }
    }
    const size_t prog_bytes = (size_t)(low_len + 4 * S + comb.size() + 4) * (v2 ? 16 : 12) + (sliced ? S * 8 : 0), col_bytes = (col_tab.size() ? col_tab.size() : 1) * 8;
    const size_t const_bytes = (size_t)(nc_all ? nc_all : 1) * sizeof(QC29) * 2, tev_bytes = tev.size() * sizeof(Fr);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    char* d = (char*)ctx->get_scratch(SC_POLY, al(prog_bytes) + al(col_bytes) + al(const_bytes) + al(tev_bytes) + 256);
    if (!d) return ZK_ERR_OOM;
    uint32_t* d_prog = (uint32_t*)d;
    const Fr** d_cols = (const Fr**)(d + al(prog_bytes));
    QC29* d_consts = (QC29*)(d + al(prog_bytes) + al(col_bytes));
    Fr* d_tev = (Fr*)(d + al(prog_bytes) + al(col_bytes) + al(const_bytes));
    // program, column table, constants (R and R' forms) and vanishing inverses travel as ONE upload: the proof makes hundreds of
    // these calls (compressions, linear combinations, class programs), and five small copies each were 2 500 copies per
    // SuperCircuit-shape proof at ~8 us of device time apiece
    const size_t total_bytes = al(prog_bytes) + al(col_bytes) + al(const_bytes) + al(tev_bytes);
    std::vector<char> staging(total_bytes, 0);
    std::vector<uint32_t> slice_tab(2 * S);
    size_t slice_tab_at = 0, comb_at = 0;
    {
        uint32_t* hp = (uint32_t*)staging.data();
        size_t w = 0;
        for (uint32_t sl = 0; sl < S; ++sl) {
            slice_tab[2 * sl] = (uint32_t)w; slice_tab[2 * sl + 1] = (uint32_t)lows[sl].size() + 1;
            for (const LowInstr& in : lows[sl]) { hp[w++] = in.w0; hp[w++] = in.a; hp[w++] = in.b; if (v2) hp[w++] = q_class_mask(in.w0); }
            for (int e = 0; e < 4; ++e) { hp[w++] = Q_END; hp[w++] = 0; hp[w++] = 0; if (v2) hp[w++] = 0; }      // END + the instructions the kernel fetches ahead
        }
        comb_at = w;
        if (sliced) {
            for (const LowInstr& in : comb) { hp[w++] = in.w0; hp[w++] = in.a; hp[w++] = in.b; hp[w++] = q_class_mask(in.w0); }
            for (int e = 0; e < 4; ++e) { hp[w++] = Q_END; hp[w++] = 0; hp[w++] = 0; hp[w++] = 0; }
        }
        slice_tab_at = w;
        if (sliced) for (uint32_t q = 0; q < 2 * S; ++q) hp[w++] = slice_tab[q];
        if (!col_tab.empty()) memcpy(staging.data() + al(prog_bytes), col_tab.data(), col_tab.size() * 8);
        char* hc = staging.data() + al(prog_bytes) + al(col_bytes);
        QC29* hq = (QC29*)hc;                                  // limb form: the R-form constants, then their R' images
        for (uint32_t j = 0; j < nc_all; ++j) {
            const Q29 a = unpack29<Fr29P>(consts_r[j]), b = unpack29<Fr29P>(consts_rp[j]);
            for (int q = 0; q < 9; ++q) { hq[j].l[q] = a.l[q]; hq[nc_all + j].l[q] = b.l[q]; }
        }
        if (!tev.empty()) memcpy(staging.data() + al(prog_bytes) + al(col_bytes) + al(const_bytes), tev.data(), tev_bytes);
    }
    ZK_HIP(ctx, hipMemcpyAsync(d, staging.data(), total_bytes, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // staging vectors are stack-owned
    size_t lds = (size_t)(v2 ? (depth > 1 ? depth - 1 : 1) : (depth > 2 ? depth - 2 : 1)) * 9 * Q_THREADS * 4;    // the topmost element(s) are in registers
    {   // measurement knob: pad the workgroup's LDS to this many bytes, i.e. cap the workgroups resident per CU (160 KB / pad)
        static const long pad = getenv("ZK_QUOTIENT_LDS_PAD") ? atol(getenv("ZK_QUOTIENT_LDS_PAD")) : 0;
        if (pad > 0 && (size_t)pad > lds && (size_t)pad <= (size_t)Q_MAX_STACK * 9 * Q_THREADS * 4) lds = (size_t)pad;
    }
    if (!ctx->quotient_attr_set) {
        const int lds_max = Q_MAX_STACK * 9 * Q_THREADS * 4;
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_quotient_eval<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_quotient_eval<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_quotient_eval<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_quotient_eval<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_quotient_eval2<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_quotient_eval2<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_quotient_eval2<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_quotient_eval2<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        ctx->quotient_attr_set = true;
    }
    ZkProfScope ps(ctx, ctx->prof_tag ? ctx->prof_tag : "quotient_eval");
    if (ctx->prof_on) {
        // algorithmic bytes of this launch: every distinct (column, rotation) operand is read once per row, parked
        // intermediates are written and read back once each, one result per row is written
        std::vector<uint64_t> ops;
        uint64_t tmp_moves = 0;
        for (uint32_t i = 0; i < num_instr; ++i) {
            const uint32_t op = h_program[3 * i];
            if (op == Q_PUSH_COL) ops.push_back(((uint64_t)h_program[3 * i + 1] << 32) | h_program[3 * i + 2]);
            else if (op == Q_TEE_TMP || op == Q_PUSH_TMP) ++tmp_moves;
        }
        std::sort(ops.begin(), ops.end());
        ops.erase(std::unique(ops.begin(), ops.end()), ops.end());
        ps.bytes = (ops.size() + tmp_moves + 1) * ne * 32;
    }
    {
        const dim3 grid((unsigned)((ne + Q_THREADS - 1) / Q_THREADS) * S), block(Q_THREADS);
        const bool full = ne >= (uint64_t)Q_THREADS;
        auto launch = [&](auto kern) {
            hipLaunchKernelGGL(kern, grid, block, lds, ctx->stream, (const uint32_t*)d_prog, low_len + 1, (const Fr* const*)d_cols, (const QC29*)d_consts, (const QC29*)(d_consts + nc_all),
                               tev.empty() ? (const Fr*)nullptr : (const Fr*)d_tev, ext_k, k, (Fr*)d_out, d_tmp, d_acc);
        };
        // measurement knob (results WRONG): 2^a consecutive workgroups of an XCD evaluate the same row tile -- the operand working set of the resident waves shrinks by
        // 2^a while every wave does what it did: what the launch would take if the slices of a sliced program shared their rows' operands through the L2
        static const uint32_t tile_alias = getenv("ZK_QUOTIENT_TILE_ALIAS") ? (uint32_t)atoi(getenv("ZK_QUOTIENT_TILE_ALIAS")) : 0;
        auto launch2 = [&](auto kern) {
            hipLaunchKernelGGL(kern, grid, block, lds, ctx->stream, (const uint32_t*)d_prog, low_len + 1, (const Fr* const*)d_cols, (const QC29*)d_consts, (const QC29*)(d_consts + nc_all),
                               tev.empty() || sliced ? (const Fr*)nullptr : (const Fr*)d_tev, ext_k, k, sliced ? d_part : (Fr*)d_out, d_tmp, d_acc, sliced ? 0u : tile_alias,
                               (const uint32_t*)d_prog + slice_tab_at, sliced ? S : 0u);
        };
        if (v2 && full && acc_in_mem) launch2(k_quotient_eval2<true, true>);
        else if (v2 && full) launch2(k_quotient_eval2<true, false>);
        else if (v2 && acc_in_mem) launch2(k_quotient_eval2<false, true>);
        else if (v2) launch2(k_quotient_eval2<false, false>);
        else if (full && acc_in_mem) launch(k_quotient_eval<true, true>);
        else if (full) launch(k_quotient_eval<true, false>);
        else if (acc_in_mem) launch(k_quotient_eval<false, true>);
        else launch(k_quotient_eval<false, false>);
    }
    ZK_CHECK_LAUNCH(ctx);
    if (sliced) {
        const dim3 grid((unsigned)((ne + Q_THREADS - 1) / Q_THREADS)), block(Q_THREADS);
        hipLaunchKernelGGL((k_quotient_eval2<true, false>), grid, block, (size_t)9 * Q_THREADS * 4, ctx->stream, (const uint32_t*)d_prog + comb_at, S + 1, (const Fr* const*)d_cols, (const QC29*)d_consts,
                           (const QC29*)(d_consts + nc_all), tev.empty() ? (const Fr*)nullptr : (const Fr*)d_tev, ext_k, k, (Fr*)d_out, d_tmp, (uint32_t*)nullptr, 0u, (const uint32_t*)nullptr, 0u);
        ZK_CHECK_LAUNCH(ctx);
    }
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
