/*
 * zkmi355 -- C ABI of the MI355X-native Halo2/KZG proving hot path (BN254).
 *
 * This is the drop-in boundary of SURVEY.md section 8(b): a thin Rust shim crate that keeps the
 * `halo2_proofs` API (create_proof / keygen_* / ParamsKZG, transcript + RNG + Circuit::synthesize
 * stay in Rust) binds exactly these symbols; see INTEGRATION.md for the `extern "C"` block.
 * The reference has no FFI for this path today; the interfaces each entry point replaces live in
 * the external crates pinned by the reference (`[REF Cargo.lock:2214-2216]` halo2_proofs 1.1.0 @
 * scroll-tech/halo2 e5ddf67, `[REF Cargo.lock:2239-2241]` halo2curves 0.1.0 @ a495a7b) and are
 * reached from the reference call sites listed per function below.
 *
 * Conventions
 *   - every function returns 0 (ZK_OK) or a negative zk_status; nothing unwinds across the ABI;
 *     zk_last_error(ctx) returns a human-readable message for the last failure on that ctx.
 *   - Fr / Fq element: 32 bytes = 4 x u64 little-endian limbs, MONTGOMERY form (R = 2^256):
 *     byte-identical to halo2curves' in-memory layout and to SerdeFormat::RawBytes.
 *   - G1Affine: 64 bytes {x, y}; identity = all zero.   G1 (Jacobian): 96 bytes {x, y, z}.
 *   - pointers named d_* are DEVICE pointers (from zk_buf_alloc or any hipMalloc / torch tensor);
 *     pointers named h_* are host pointers.  The caller owns every buffer it passes.
 *   - a zk_ctx is used by one host thread at a time; different contexts are independent.
 *   - all device work is enqueued on the context's stream (zk_ctx_set_stream); functions that
 *     return host results synchronise that stream before returning.
 */
#ifndef ZKMI355_H
#define ZKMI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zk_ctx zk_ctx;
typedef struct zk_srs zk_srs;

typedef enum zk_status {
    ZK_OK = 0,
    ZK_ERR_INVALID_ARG = -1,  /* null pointer, n not a power of two, k > 28 (Fr two-adicity) ... */
    ZK_ERR_HIP = -2,          /* a HIP runtime call failed (message has the HIP error string)   */
    ZK_ERR_OOM = -3,          /* device allocation failed                                       */
    ZK_ERR_NO_DEVICE = -4,    /* no gfx950 device visible: there is NO CPU fallback            */
    ZK_ERR_UNSUPPORTED = -5
} zk_status;

enum { ZK_FIELD_FR = 0, ZK_FIELD_FQ = 1 };
enum { ZK_OP_ADD = 0, ZK_OP_SUB = 1, ZK_OP_MUL = 2 };

/* ---- context / memory ----------------------------------------------------------------------- */
int zk_ctx_create(int device, zk_ctx** out);
void zk_ctx_destroy(zk_ctx* ctx);
const char* zk_last_error(const zk_ctx* ctx);
/* Use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
int zk_ctx_set_stream(zk_ctx* ctx, void* hip_stream);
/* Waits for EVERYTHING this context has enqueued: the library runs on six HIP streams (main, copy, auxiliary transforms,
 * three MSM side streams) and zk_ctx_sync joins all of them, not only the main one -- entry points that return host data
 * (commitments, proofs, downloads) are complete when they return; asynchronous ones (zk_ntt, zk_vec_*, zk_coeff_to_coset ...)
 * are complete after zk_ctx_sync.  A caller-supplied main stream (zk_ctx_set_stream) is synchronised as well. */
int zk_ctx_sync(zk_ctx* ctx);
/* Diagnostics for the contract above: the number of library streams with work still in flight (0 after zk_ctx_sync), and
 * a delay of about `usec` microseconds on one stream (role 0 main, 1 copy, 2 auxiliary, 3..5 MSM side streams). */
int zk_ctx_streams_busy(zk_ctx* ctx, int* busy);
int zk_ctx_debug_delay(zk_ctx* ctx, int role, uint32_t usec);
int zk_buf_alloc(zk_ctx* ctx, size_t bytes, void** d_ptr);
int zk_buf_free(zk_ctx* ctx, void* d_ptr);
/* Page-locked host memory for witness columns: uploads from it run at PCIe speed and overlap with
 * compute (zk_commit_batch_h2d, zk_proof_advice_phase); pageable memory works too, at about half. */
int zk_host_alloc(zk_ctx* ctx, size_t bytes, void** h_ptr);
int zk_host_free(zk_ctx* ctx, void* h_ptr);
/* Page-lock memory the caller already owns (e.g. the Vec<Fr> columns of a witness that is reused
 * across proofs) instead of copying it into zk_host_alloc memory; unregister before freeing it.   */
int zk_host_register(zk_ctx* ctx, void* h_ptr, size_t bytes);
int zk_host_unregister(zk_ctx* ctx, void* h_ptr);
int zk_h2d(zk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int zk_d2h(zk_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int zk_d2d(zk_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
/* HIP-event timing on the context's stream (bench.py's roofline leg) */
int zk_timer_start(zk_ctx* ctx);
int zk_timer_stop_ms(zk_ctx* ctx, float* ms);   /* synchronises on the stop event */

/* Per-kernel HIP-event profiling on the context stream.  While enabled, every named kernel group
 * ("msm_digits", "msm_sort", "msm_buckets", "msm_reduce", "msm_tail_host", "ntt_pass", "ntt_last",
 * ...) is bracketed by an event pair; zk_prof_get drains the stream and returns the accumulated
 * device milliseconds and launch count for one name.  on = 2 records only the groups of the roofline
 * kernels ("msm_buckets", "ntt_*", "quotient*"): every event pair costs a little stream time, and a
 * throughput measurement should carry as few as it needs.                                        */
int zk_prof_enable(zk_ctx* ctx, int on);
int zk_prof_reset(zk_ctx* ctx);
int zk_prof_get(zk_ctx* ctx, const char* name, double* total_ms, uint64_t* count);
/* algorithmic HBM bytes of the launches booked under `name` (scopes that state them: the expression evaluator
 * counts 32 B per row for every distinct (column, rotation) operand, parked intermediate and result); 0 otherwise */
int zk_prof_get_bytes(zk_ctx* ctx, const char* name, uint64_t* bytes);
/* writes a ';'-separated list of the names seen so far into buf */
int zk_prof_names(zk_ctx* ctx, char* buf, size_t len);

/* Self-test of the device's Montgomery products (no reference counterpart: halo2curves' `Fr::mul` has one form; here the device runs
 * generated gfx950 asm -- csrc/mul29_asm.hip.hpp -- and the host / the definition the C forms of csrc/ff29.hip.hpp): `operand_sets`
 * lanes each run mul29 / mul29_ub / sqr29 / mul2add29 in both forms on pseudo-random operands AT the documented lazy-reduction
 * bounds and compare every limb.  out3 = {lanes with a difference, OR of the differing routines (1, 2, 4, 8), lanes run}. */
int zk_selftest_products(zk_ctx* ctx, int field, uint32_t operand_sets, uint32_t seed, uint32_t* out3);

/* ---- field vectors (halo2curves Fr/Fq Add/Sub/Mul, element-wise)  -- SURVEY 8a K4/K10 ---------- */
int zk_field_vec_op(zk_ctx* ctx, int field, int op, const void* d_a, const void* d_b, void* d_out, size_t n);
/* out[i] = a[i] * s  (s: host pointer to one element) */
int zk_fr_scale(zk_ctx* ctx, void* d_a, const void* h_s, size_t n);
/* ff::BatchInvert semantics (zeros stay zero), in place -- SURVEY 8a K12 */
int zk_fr_batch_invert(zk_ctx* ctx, void* d_a, size_t n);

/* ---- NTT: halo2_proofs::arithmetic::best_fft / poly::EvaluationDomain  -- SURVEY 8a K2, K3 ---- */
/* In-place size-2^log_n transform over <omega>, natural order in and out:
 *   out[i] = sum_j a[j] * omega^(i*j),   omega = ROOT_OF_UNITY^(2^(28-log_n))   (inverse = 0)
 *   inverse = 1: omega^-1 and the 1/n scale, i.e. EvaluationDomain::lagrange_to_coeff.          */
int zk_ntt(zk_ctx* ctx, void* d_data, uint32_t log_n, int inverse);
/* zk_ntt over `count` columns of the same size, in place, several columns per launch. */
int zk_ntt_batch(zk_ctx* ctx, void* const* d_datas, size_t count, uint32_t log_n, int inverse);
/* best_fft with an arbitrary primitive 2^log_n-th root (h_omega: one Fr), no scaling.            */
int zk_ntt_omega(zk_ctx* ctx, void* d_data, uint32_t log_n, const void* h_omega);
/* ONE transform of size 2^log_n spread over `world` contexts, one per GPU (the
 * 4-step split with a single all-to-all).  On entry rank r holds the residue class
 * x[r + world * i], i < m = 2^log_n / world, in d_local; on return d_local[j1 * (m / world) + c]
 * = X[(r * (m / world) + c) + m * j1]: every rank owns the outputs whose index mod m falls in its
 * slice.  exchange(user, d_send, bytes_per_peer, d_recv) is an all-to-all of DEVICE buffers:
 * block p of d_send goes to rank p, block p of d_recv comes from rank p (RCCL all_to_all_single
 * over xGMI, see zkevm-circuits_amd/sharding.py); it must return 0 with d_recv complete.
 * world: power of two <= 16, 2^log_n >= world^2.  inverse = 1 applies omega^-1 and 1/n.
 * exchange == NULL: the context's own RCCL communicator does the all-to-all (zk_comm_init), with
 * no host synchronisation between the local transform, the exchange and the cross butterflies.
 * Worth it only for transforms far above 2^24: a 2^20 NTT takes 0.12 ms on one GPU.             */
typedef int (*zk_alltoall_fn)(void* user, const void* d_send, size_t bytes_per_peer, void* d_recv);
int zk_ntt_sharded(zk_ctx* ctx, void* d_local, uint32_t log_n, int inverse, uint32_t rank, uint32_t world, zk_alltoall_fn exchange, void* user);
/* EvaluationDomain::coeff_to_extended: d_coeffs (2^k) -> d_out (2^ext_k): scale by zeta^i,
 * zero-pad, NTT over the extended domain.                                                       */
int zk_coeff_to_extended(zk_ctx* ctx, const void* d_coeffs, uint32_t k, uint32_t ext_k, void* d_out);
/* One coset of the extended domain (EvaluationDomain::coeff_to_extended_part): d_out[i] =
 * f(g * omega^i) for i < 2^k, f given by 2^k coefficients; h_g is one Fr.  The extended domain of
 * halo2 is the union of the cosets g_r = zeta * omega_ext^r, r < 2^(ext_k-k): extended index
 * i * 2^(ext_k-k) + r  <->  element i of coset r.  d_out may alias d_coeffs.                       */
int zk_coeff_to_coset(zk_ctx* ctx, const void* d_coeffs, uint32_t k, const void* h_g, void* d_out);
/* The same for `count` polynomials on one coset, several columns per launch (the quotient reads hundreds of
 * columns on every coset; below 2^20 rows one column does not fill the device). */
int zk_coeff_to_coset_batch(zk_ctx* ctx, const void* const* d_coeffs, uint32_t k, const void* h_g, void* const* d_outs, size_t count);
/* d_dst[i * stride + offset] = d_src[i] * scale, i < n: interleaves coset values into extended order */
int zk_fr_scatter_scaled(zk_ctx* ctx, const void* d_src, size_t n, const void* h_scale, void* d_dst, size_t stride, size_t offset);
/* EvaluationDomain::extended_to_coeff: inverse NTT over the extended domain, unscale by zeta^-i;
 * in place on d_ext (2^ext_k); the caller truncates to n*(deg-1).                               */
int zk_extended_to_coeff(zk_ctx* ctx, void* d_ext, uint32_t ext_k);

/* ---- polynomial helpers: halo2_proofs::arithmetic  -- SURVEY 8a K9, K10 ------------------------ */
/* eval_polynomial(coeffs, x) -> one Fr on the host */
int zk_poly_eval(zk_ctx* ctx, const void* d_coeffs, size_t n, const void* h_x, void* h_out);
/* eval_polynomial of `count` polynomials (device pointers, n coefficients each) at one point:
 * one power table and one synchronisation for all of them; h_out receives count Fr               */
int zk_poly_eval_batch(zk_ctx* ctx, const void* const* d_coeff_ptrs, size_t count, size_t n, const void* h_x, void* h_out);
/* eval_polynomial for (polynomial, point) pairs in one pass: h_out[j] = d_coeff_ptrs[j](h_points[point_index[j]]).  The evaluations
 * of a proof (plonk/prover.rs: every advice / fixed / permutation / lookup query at x w^rot) open a dozen distinct points when a
 * column is read at many rotations; this is one table build, one Horner launch and one download for all of them.              */
int zk_poly_eval_pairs(zk_ctx* ctx, const void* const* d_coeff_ptrs, const uint32_t* point_index, size_t count, const void* h_points, size_t num_points, size_t n, void* h_out);
/* kate_division(coeffs, z): d_q receives n-1 coefficients of (f(X) - f(z)) / (X - z)             */
int zk_kate_division(zk_ctx* ctx, const void* d_coeffs, size_t n, const void* h_z, void* d_q);
/* z[0] = 1 (product) / 0 (sum); z[i+1] = z[i] (*|+) a[i]: the permutation / lookup grand product
 * and grand sum.  d_z may not alias d_a.                                      */
int zk_fr_prefix_product(zk_ctx* ctx, const void* d_a, void* d_z, size_t n);
int zk_fr_prefix_sum(zk_ctx* ctx, const void* d_a, void* d_z, size_t n);

/* ---- quotient / expression evaluation: halo2_proofs::plonk::evaluation (evaluate_h, GraphEvaluator)
 * + vanishing divide_by_vanishing_poly  -- SURVEY 8a K4, K5 ------------------------------------ */
/* Runs one postfix program per row i of a 2^ext_k domain and writes out[i] (see quotient.hip for
 * the instruction set: 3 x u32 per instruction {op, a, b}).  Columns are device pointers to
 * 2^ext_k Fr values each (extended-coset evaluations, or Lagrange values when ext_k == k);
 * PUSH_COL reads row (i + rot * 2^(ext_k-k)) mod 2^ext_k.  divide_by_vanishing != 0 multiplies the
 * result by 1/(X^n - 1) evaluated on the zeta-coset.  d_out must not alias any input column.
 * TEE_TMP t / PUSH_TMP t keep GraphEvaluator-style intermediates of the row in device scratch
 * (t < 4096, 2^ext_k x 32 B each); reading one before it is written is ZK_ERR_INVALID_ARG.          */
int zk_quotient_eval(zk_ctx* ctx, const uint32_t* h_program, uint32_t num_instr, const void* const* h_col_ptrs, uint32_t num_cols,
                     const void* h_consts, uint32_t num_consts, uint32_t k, uint32_t ext_k, int divide_by_vanishing, void* d_out);
/* Host only (no device): the instruction stream the evaluator's kernel runs for `program` -- memory operands fused
 * into the operations (ADD_COL 16, SUB_COL 17, RSUB_COL 18, MUL_COL 19, FOLD_COL 20 | const << 12, NOP 21), bit 8 / 9
 * of word 0 = "settle t0 / t1 first", parked intermediates as columns num_cols + slot.  fuse = 0 keeps the caller's
 * sequence.  out_words may be NULL (sizes only).  For tests and for inspecting what a key's gates compile to. */
int zk_host_quotient_lower(const uint32_t* program, uint32_t num_instr, uint32_t num_cols, int fuse, uint32_t* out_words, size_t cap_words,
                           uint32_t* out_instr, int* out_depth);
/* Host only (no device): where zk_quotient_eval would cut `program` into slices for 2^ext_k rows (round 6: a large sum of terms
 * acc = acc * c_f + term_f whose operands are read many times is evaluated slice by slice over the same rows at the same time so
 * that the slices find each other's operands in the caches; cuts only at top-level FOLDs that no parked intermediate is alive
 * across).  out_cuts receives the first instruction of every slice and the end of the program (*out_count entries; 0 = the program
 * runs in one piece).  ZK_QUOTIENT_SLICES=0 / N turns the slicing off / asks for N slices.  No counterpart in halo2's evaluate_h
 * (plonk/evaluation.rs, external crate), which walks one row at a time on the CPU.                                              */
int zk_host_quotient_slices(const uint32_t* program, uint32_t num_instr, uint32_t ext_k, uint32_t* out_cuts, size_t cap, uint32_t* out_count);

/* Host only, for tests: how the prover distributes constraint programs over its degree classes when the programs share
 * intermediates (TEE_TMP in one constraint, PUSH_TMP in a later one -- halo2's GraphEvaluator intermediates as exported): a
 * class that reads an intermediate it has not computed re-materialises the defining sub-expression.  words: 3 per instruction,
 * `count` programs back to back; cls[i] < classes.  Output: the class programs back to back, each constraint followed by the
 * marker {FOLD (9), i, 0}.  *conflict = 1 for slot reuse across constraints (the prover then evaluates a single class). */
int zk_host_split_programs(const uint32_t* words, const uint32_t* lens, const uint32_t* cls, uint32_t count, uint32_t classes,
                           uint32_t* out_words, size_t out_cap_words, uint32_t* out_lens, int* conflict);
/* Host only, for tests: the additive split of ONE constraint program over the degree classes 0 .. E as the prover applies it: a
 * constraint that is a sum of terms of different degrees (q (a b - c): degree 3 and degree 2) hands each term to the class of its
 * own degree -- h is linear in the constraints -- so that columns occurring only in low-degree terms are transformed to fewer cosets
 * (halo2's evaluate_h, external crate, evaluates every constraint on the whole extended domain; same h: the prover carries what a
 * moved term is on H along as a remainder polynomial, so that every class stays divisible by X^n - 1).  Output: the pieces back to
 * back, out_lens[j] instructions of class out_cls[j]; one piece (the program itself) when it is not split.  out_words may be NULL
 * (count only). */
int zk_host_additive_split(const uint32_t* words, uint32_t num_instr, uint32_t E, uint32_t* out_words, size_t out_cap_words, uint32_t* out_cls, uint32_t* out_lens,
                           uint32_t cap_pieces, uint32_t* num_pieces);
/* Host only, for tests: the program of ONE degree class as the prover assembles it from the class's terms (term t belongs to
 * constraint cons[t] < K of the K constraints halo2's evaluate_h folds with y, external crate): the weighted sum
 * sum_t y^(K-1-cons[t]) term_t with the terms that share a single-column factor -- a selector, l_active -- collected under ONE product by
 * that factor, instead of folding the terms one by one (a product per term for the folding, another for its selector).  Same
 * polynomial, same h.  A weight is the constant 0xFFFC0000 + (K-1-cons[t]) (= that power of y).  Terms that park or read
 * intermediates keep their definitions ahead of their readers.  ZK_ERR_UNSUPPORTED when the prover would keep the folded form (a
 * stack deeper than the evaluator's).  out_words may be NULL (count only). */
int zk_host_group_terms(const uint32_t* words, const uint32_t* lens, const uint32_t* cons, uint32_t count, uint32_t K, uint32_t* out_words, size_t out_cap_words,
                        uint32_t* out_instr);

/* Host only, for tests: the program of ONE degree class as the prover COMPILES it (round 6; csrc/class_compile.hpp) -- what halo2's
 * GraphEvaluator (plonk/evaluation.rs, external crate) does for a circuit whose gates share sub-expressions, done on this side of the
 * boundary: the terms become one hash-consed expression graph (TEE_TMP / PUSH_TMP of the incoming programs resolved), the y-weighted
 * sum is regrouped under common FACTORS of any shape (a selector product q_usable * q_step * state_selector, a gadget's condition
 * [REF zkevm-circuits/src/evm_circuit/execution.rs:832-851]), sums run in Horner form over the constraint index, values still used
 * twice are parked in slots assigned by liveness.  The program leaves acc = sum_t y^(*out_last - cons[t]) term_t.  out_stats[0..5] =
 * graph nodes, values parked, slots alive at once, products, factor groups, stack depth.  ZK_ERR_UNSUPPORTED: stack too deep (the
 * prover then keeps zk_host_group_terms' form). */
int zk_host_compile_class(const uint32_t* words, const uint32_t* lens, const uint32_t* cons, uint32_t count, uint32_t K, uint32_t* out_words, size_t out_cap_words,
                          uint32_t* out_instr, uint32_t* out_last, uint32_t* out_stats);
/* Host only (no device, no SRS): the quotient plan of a constraint system -- degree classes, additive split, each class's compiled
 * program -- exactly as zk_proof_finish will follow it; cs_blob = the constraint-system part of a zk_pk_create blob (column data
 * not needed).  out_summary: [0] ext_k - k, [1] constraints, [2] classes on, [3] additive split on, [4] expression graph on,
 * [5] remainder polynomials, [6] cost estimate, [7] columns read; then 8 words per class: used, instructions, products, columns,
 * values parked, slots alive at once, factor groups, last.  class_index / out_words / out_instr: one class's program (optional). */
int zk_host_quotient_plan(const void* cs_blob, size_t blob_len, uint32_t* out_summary, size_t cap_summary, uint32_t class_index, uint32_t* out_words, size_t out_cap_words,
                          uint32_t* out_instr);

/* out[i] = base^i * mul for i < n (Montgomery form): omega-power / delta-power "columns"          */
int zk_fr_powers(zk_ctx* ctx, const void* h_base, const void* h_mul, void* d_out, size_t n);
/* logUp multiplicities (halo2 Scroll fork, plonk/mv_lookup/prover.rs: m(X)): d_m[i] = number of rows
 * r < usable_rows with d_inputs[r] == d_table[i] for i < usable_rows (a value that occurs in several
 * table rows is credited to the lowest one), 0 for usable_rows <= i < n.  *bad_row = lowest input
 * row whose value is not in the table, UINT64_MAX when every input is found.                        */
int zk_lookup_multiplicities(zk_ctx* ctx, const void* d_inputs, const void* d_table, size_t usable_rows, void* d_m, size_t n, uint64_t* bad_row);
/* n uniformly random Fr (Montgomery form) on the device: element i = Fr::from_uniform_bytes of
 * ChaCha20 block (first_block + i) under key32 / stream_id (64-bit counter in state words 12..13,
 * stream id in 14..15).  What the prover draws its blinding polynomial from (halo2 takes an RngCore
 * from the caller; plonk/vanishing/prover.rs fills the random poly from it element by element).    */
int zk_fr_random(zk_ctx* ctx, const uint8_t* key32, uint64_t stream_id, uint64_t first_block, void* d_out, size_t n);

/* ---- SRS: halo2_proofs::poly::kzg::commitment::ParamsKZG  -- SURVEY 8a A5 ---------------------- */
/* Upload g (n = 2^k G1Affine) and g_lagrange; host pointers.  h_g_lagrange may be NULL
 * (ParamsKZG::from_parts with None): the Lagrange basis is then derived on the device.            */
int zk_srs_create(zk_ctx* ctx, uint32_t k, const void* h_g, const void* h_g_lagrange, zk_srs** out);
/* ParamsKZG::unsafe_setup_with_s(k, s): g[i] = s^i * G, g_lagrange[i] = L_i(s) * G, on device.    */
int zk_srs_setup_with_s(zk_ctx* ctx, uint32_t k, const void* h_s, zk_srs** out);
/* ParamsKZG::downsize (prover/src/common/prover.rs:40-60: every layer shrinks the one params file
 * to its own degree): the first 2^new_k points of g, with the Lagrange basis of the smaller domain
 * recomputed on the device (g_to_lagrange: an inverse FFT over G1).                                 */
int zk_srs_downsize(zk_ctx* ctx, const zk_srs* srs, uint32_t new_k, zk_srs** out);
/* ---- SRS files: ParamsKZG::{read_custom, write_custom}; the reference loads `params{k}` through
 * prover::utils::load_params [REF prover/src/utils.rs:39-84], default format RawBytesUnchecked
 * [REF prover/src/utils.rs:32].  file = u32 k (LE) | g[n] | g_lagrange[n] | g2 | s_g2; G1 is 64 B
 * (formats 1, 2: the in-memory Montgomery image) or 32 B (format 0: x canonical LE, bit 254 = parity
 * of y, bit 255 = identity -- halo2curves @ a495a7b, pinned by the vk / proof bytes of
 * [REF aggregator/data/batch-task.json]); G2 is twice that and is handed through untouched.        */
#define ZK_SERDE_PROCESSED 0
#define ZK_SERDE_RAW 1            /* RawBytes: every point is checked (limbs < p, on the curve)      */
#define ZK_SERDE_RAW_UNCHECKED 2
/* 4 + 2 * 2^k * g1 + 2 * g2 bytes, the length load_params insists on; 0 for a bad k / format       */
size_t zk_params_file_len(uint32_t k, int format);
/* h_file: the whole file in host memory.  A length that does not match the k in its header is
 * refused before anything is parsed, like the reference.  h_g2 / h_s_g2 (nullable) receive the two
 * G2 encodings as they are in the file (128 B each, 64 B for Processed).                           */
int zk_params_read(zk_ctx* ctx, const void* h_file, size_t len, int format, zk_srs** out, void* h_g2, void* h_s_g2);
/* h_out = NULL: only *len is written (size query).  h_g2 / h_s_g2: the encodings to put in the file */
int zk_params_write(zk_ctx* ctx, const zk_srs* srs, const void* h_g2, const void* h_s_g2, int format, void* h_out, size_t cap, size_t* len);
/* unsafe_setup_with_s, G2 half (host only): the generator and s * generator as RawBytes
 * (x.c0 | x.c1 | y.c0 | y.c1, Montgomery limbs; 128 B each); h_s: one Montgomery Fr                */
int zk_g2_setup(const void* h_s, void* h_g2, void* h_s_g2);
void zk_srs_destroy(zk_ctx* ctx, zk_srs* srs);
uint32_t zk_srs_k(const zk_srs* srs);
const void* zk_srs_g(const zk_srs* srs);          /* device pointer, n G1Affine */
const void* zk_srs_g_lagrange(const zk_srs* srs); /* device pointer or NULL     */

/* ---- MSM: halo2_proofs::arithmetic::best_multiexp  -- SURVEY 8a K1 ----------------------------- */
/* sum_i scalars[i] * bases[i] over device buffers; the affine result (64 B) is written to the
 * host.  Scalars are Montgomery-form Fr exactly as halo2 holds them.                             */
int zk_msm_g1(zk_ctx* ctx, const void* d_scalars, const void* d_bases, size_t n, void* h_out_affine);
/* ParamsKZG::commit (basis = 0, over g) / commit_lagrange (basis = 1, over g_lagrange).          */
int zk_commit(zk_ctx* ctx, const zk_srs* srs, int basis, const void* d_scalars, size_t n, void* h_out_affine);
/* `count` commitments over the same basis (ParamsKZG::commit_lagrange over every advice column
 * of a phase): d_scalar_ptrs[i] addresses n Fr on the device, h_out_affine receives count x 64 B.
 * Consecutive MSMs are pipelined on side streams.  Each column is judged on the device first (4096
 * sampled cells): columns with at most a quarter of field-sized cells (>= 2^64) take the per-window
 * path of zk_commit_batch_hint's hint 1 -- a choice that affects speed only, never the result.     */
int zk_commit_batch(zk_ctx* ctx, const zk_srs* srs, int basis, const void* const* d_scalar_ptrs, size_t count, size_t n, void* h_out_affine);
/* zk_commit_batch for columns still in host memory: column i+1 is uploaded (copy stream) while the
 * MSM of column i runs; d_cols[i] (n x 32 B device buffers) receive the columns.  This is the shape
 * of halo2's advice commitment loop (plonk/prover.rs: one commit_lagrange per witness column).     */
/* zk_commit_batch with a hint per column (nullable array).  narrow[i] = 1: column i holds small
 * integers (below 2^64: selectors, bytes, counters, lookup multiplicities), whose digits leave most
 * Pippenger windows empty; such a column takes the per-window MSM path, which skips empty windows,
 * instead of the merged-window path, which always pays for its 2^(c-1) shared buckets.
 * narrow[i] = 2: dense values in long runs of equal scalars (running products / sums that stay
 * constant over stretches of rows: permutation products of a circuit with few copy constraints).  A
 * column of at least 4096 rows with at most n / 16 runs is committed by its run ends -- Abel
 * summation against a prefix-sum table of the basis built on first use (64 B per SRS point,
 * csrc/runs.hip; ZK_MSM_RUNS=0 turns that off); otherwise the merged-window path with the sliced
 * bucket sort, whose four workgroups per partition stream a run-filled partition faster than the
 * one-launch sort does.
 * narrow[i] = 3: a running sum whose increments are mostly equal (a lookup's phi: the increment is
 * the same on every row where the lookup is switched off and the table row unused).  A full-length
 * column over the Lagrange basis is committed as an MSM of s_j = c - (z_{j+1} - z_j) -- zero wherever
 * the increment is the common value c -- over the prefix sums of the basis, plus c times a fixed
 * point (csrc/runs.hip; ZK_MSM_DIFF=0 turns that off); a column whose increments are not mostly
 * equal costs what a dense column costs.
 * 0: dense.  The hint affects speed only.  zk_commit_batch_h2d and zk_proof_advice_phase derive
 * 0 / 1 themselves from a sample of the host column.                                              */
int zk_commit_batch_hint(zk_ctx* ctx, const zk_srs* srs, int basis, const void* const* d_scalar_ptrs, size_t count, size_t n, const uint8_t* narrow, void* h_out_affine);
int zk_commit_batch_h2d(zk_ctx* ctx, const zk_srs* srs, int basis, const void* const* h_cols, void* const* d_cols, size_t count, size_t n, void* h_out_affine);
/* best_multiexp over HOST slices, exactly the reference signature (copies in, computes, copies
 * the affine result out).                                                                        */
int zk_msm_g1_host(zk_ctx* ctx, const void* h_scalars, const void* h_bases, size_t n, void* h_out_affine);
/* Introspection: Pippenger window size (bits) and window count that commitments of n points over
 * this SRS use -- with the SRS's fixed-base window tables all windows share one bucket set (c = 20,
 * 13 windows at k = 20); without them (tables beyond ZK_MSM_TABLE_GB, default 32 GiB per basis)
 * one bucket set per window (c <= 16).                                                            */
int zk_msm_plan(const zk_srs* srs, size_t n, int* window_bits, int* windows);
/* Host only: the merged-window plan for an SRS of 2^k points, with the shift applied to the top window's digit (the top window
 * holds only the leading bits of a scalar; its digit is scaled so that its entries spread over the bucket range). */
int zk_host_msm_plan(uint32_t k, int* window_bits, int* windows, int* top_shift);

/* Host-only: out = sum of n affine points (no context, no device).  Used to finish a point-sharded
 * MSM: each rank's 64-byte partial result is all-gathered as bytes (RCCL has no EC reduce op) and
 * summed here.                                                                                   */
int zk_g1_sum_host(const void* h_points_affine, size_t n, void* h_out_affine);

/* ---- full proof: halo2_proofs::plonk::{keygen_pk, create_proof} with the GWC or SHPLONK multi-open
 * (poly::kzg::multiopen::{ProverGWC, ProverSHPLONK})  -- SURVEY 8a A1, A4, K6-K11; csrc/prover.hip -- */
typedef struct zk_pk zk_pk;
typedef struct zk_proof zk_proof;       /* a proving session, see below */
/* keygen_pk over a flat circuit description (the "pk blob", version 3, filled by the Rust shim from
 * halo2's ConstraintSystem / by zkevm-circuits_amd/plonk.py in tests: header, phases, the advice /
 * fixed / instance query lists in registration order, permutation columns, constants, gate programs,
 * mv-lookup arguments (one table tuple + N input tuples each, as chunk_lookups() leaves them
 * [REF zkevm-circuits/src/super_circuit/test.rs:59]), fixed and sigma columns in Lagrange form;
 * layout in INTEGRATION.md).  Commits fixed and sigma columns and keeps their Lagrange and
 * coefficient forms on the device.  srs must have the circuit's k (zk_srs_downsize).  A blob whose
 * declared degree is below what its gates and lookup arguments require (halo2
 * ConstraintSystem::degree) is refused.                                                            */
int zk_pk_create(zk_ctx* ctx, const zk_srs* srs, const void* h_blob, size_t blob_len, zk_pk** out);
void zk_pk_destroy(zk_ctx* ctx, zk_pk* pk);
/* verifying-key side: (F + P) x 64-byte affine commitments (fixed, then sigma) and vk_repr (Fr)  */
int zk_pk_vk(zk_ctx* ctx, const zk_pk* pk, void* h_commitments, void* h_vk_repr);
/* `vk.transcript_repr()`: the scalar create_proof absorbs first (halo2 VerifyingKey::hash_into).
 * halo2 derives it from the Debug string of the pinned verifying key -- the value the reference
 * pins for the SuperCircuit at [REF zkevm-circuits/src/super_circuit/test.rs:70-85] -- which only
 * the Rust side can produce: the shim computes it with upstream keygen_vk and installs it here (32 B
 * Montgomery Fr), after which proofs of this key are proofs for the reference's own verify_proof.
 * Without this call the key uses a stand-in (Blake2b-512 "Halo2-Verify-Key" over the
 * constraint-system part of the blob and the compressed fixed / sigma commitments).                */
int zk_pk_set_transcript_repr(zk_ctx* ctx, zk_pk* pk, const void* h_repr_fr32);
/* out16 = k, degree, extended k, F, A, I, permutation columns, permutation chunks, lookup arguments,
 * phases, challenges, blinding factors, advice queries, fixed queries, commitments per proof,
 * evaluations per proof                                                                            */
int zk_pk_shape(zk_ctx* ctx, const zk_pk* pk, uint32_t* out16);
/* The quotient plan of a key -- how its constraints are dealt to degree classes and what each class's program costs (halo2's
 * `Evaluator::new` builds the corresponding GraphEvaluator at keygen, plonk/evaluation.rs, external crate): zk_host_quotient_plan's
 * summary (8 + 8 x (extended_k - k + 1) words) for `pk` under the measurement knobs in force.  Made on first use, kept by the key. */
int zk_pk_quotient_plan(zk_ctx* ctx, const zk_pk* pk, uint32_t* out_summary, size_t cap_summary);
/* create_proof: h_advice / h_instance are arrays of host pointers to n x 32-byte Lagrange columns;
 * seed16 seeds the XorShift blinding RNG; the proof bytes (compressed points and canonical
 * scalars, halo2 encoding) are written to h_proof.  Fails with ZK_ERR_INVALID_ARG if the witness
 * does not satisfy the copy or lookup constraints.                                               */
int zk_create_proof(zk_ctx* ctx, const zk_pk* pk, const void* const* h_advice, const void* const* h_instance, const uint8_t* seed16,
                    void* h_proof, size_t proof_cap, size_t* proof_len);

/* ---- dev::MockProver on the device: halo2_proofs::dev::MockProver::{run, verify_par, verify_at_rows_par}
 * [REF zkevm-circuits/src/test_util.rs:272], [REF prover/src/common/prover/mock.rs:18-19], [REF testool/src/statetest/executor.rs:703-714]
 * -- SURVEY 8a A9.  No commitments, no transcript: checks that the witness satisfies the key's circuit and says where it does not.
 *   ZK_MOCK_GATE         VerifyFailure::ConstraintNotSatisfied: gate polynomial `index` (position in the key's flat list of
 *                        gate polynomials = the blob's order) is not zero at `row`
 *   ZK_MOCK_LOOKUP       VerifyFailure::Lookup: input tuple `sub` of lookup argument `index` at `row` occurs in no usable row of the table
 *   ZK_MOCK_PERMUTATION  VerifyFailure::Permutation: cell `row` of permutation column `index` (position in the key's list of
 *                        permutation columns) differs from the cell sigma maps it to (sub 0); sub 1: sigma names no cell (a broken key)
 * gate_rows / lookup_rows: the row ids of verify_at_rows_par (each must be a usable row, else ZK_ERR_INVALID_ARG -- upstream
 * panics); NULL = every usable row (verify_par).  Copy constraints are always checked on all rows, as upstream does.
 * h_advice / h_instance: host pointers to n x 32-byte columns (as zk_create_proof); h_challenges: the circuit's challenges
 * (32 B each, Montgomery Fr, zk_pk_shape's count) or NULL for MockProver's own (zk_host_mock_challenges).
 * out receives the first `cap` failures sorted by (kind, index, sub, row); *count the number found (may exceed cap: then
 * which ones were kept is unspecified).  Region bookkeeping (CellNotAssigned, ConstraintPoisoned) is not modelled: the
 * witness arrives as finished columns.  Gate failures are detected with a random fold first (a satisfied witness costs one
 * evaluation pass; a failing one is missed with probability < gates / 2^253) and then listed exactly, one pass per polynomial. */
typedef struct zk_mock_failure { uint32_t kind, index, sub, row; } zk_mock_failure;
enum { ZK_MOCK_GATE = 1, ZK_MOCK_LOOKUP = 2, ZK_MOCK_PERMUTATION = 3 };
int zk_mock_verify(zk_ctx* ctx, const zk_pk* pk, const void* const* h_advice, const void* const* h_instance, const void* h_challenges,
                   const uint32_t* gate_rows, size_t num_gate_rows, const uint32_t* lookup_rows, size_t num_lookup_rows,
                   zk_mock_failure* out, size_t cap, size_t* count);
/* The same checks inside a proving session, after its last advice phase and before zk_proof_finish: over the columns the
 * session holds on the device and with the challenges its transcript produced (no second upload).  What a rejected proof
 * leaves open -- which constraint the witness breaks, and where -- in one call; the session stays usable.            */
int zk_proof_mock_verify(zk_ctx* ctx, zk_proof* proof, const uint32_t* gate_rows, size_t num_gate_rows, const uint32_t* lookup_rows, size_t num_lookup_rows,
                         zk_mock_failure* out, size_t cap, size_t* count);
/* Host only: the challenges MockProver hands a circuit (halo2 dev.rs: h = Blake2b-512("Halo2-MockProver"), then
 * challenge i = Fr::from_uniform_bytes(h = Blake2b-512(h)); the third one is the constant the reference pins at
 * [REF zkevm-circuits/src/super_circuit.rs:729]).  out: count x 32 B Montgomery Fr.                       */
int zk_host_mock_challenges(uint32_t count, void* out_fr32);

/* Phase-by-phase proving session (what the Rust shim drives: Circuit::synthesize runs on the host
 * once per phase and needs the challenges of the earlier phases -- the SuperCircuit has three
 * phases, zkevm-circuits/src/util.rs:120-133).  begin -> zk_proof_advice_phase x num_phases ->
 * finish.  zk_create_proof is begin + all phases + finish for witnesses known up front.          */
/* multi-open scheme of the session: GWC (ProverGWC, what north_star names; default) or SHPLONK
 * (ProverSHPLONK / BDFG21, what the reference's call sites instantiate: two commitments in total) */
enum { ZK_MULTIOPEN_GWC = 0, ZK_MULTIOPEN_SHPLONK = 1 };
int zk_proof_set_multiopen(zk_ctx* ctx, zk_proof* proof, int kind);
/* The vanishing argument's "random" polynomial (halo2 vanishing::Argument::commit).  ONE (default): the constant 1, commitment
 * g[0], evaluation 1 -- what the reference's own prover emitted: in [REF aggregator/data/batch-task.json: chunk_proofs[0]] that
 * commitment is (1, 2) and that evaluation is 1 (Scroll's fork commits no blinding polynomial).  UNIFORM: n uniform coefficients
 * as upstream PSE halo2 draws them (one more dense commitment).  The verifier accepts either.  Any time before zk_proof_finish.  */
enum { ZK_VANISHING_UNIFORM = 0, ZK_VANISHING_ONE = 1 };
int zk_proof_set_vanishing_random(zk_ctx* ctx, zk_proof* proof, int kind);
int zk_proof_begin(zk_ctx* ctx, const zk_pk* pk, const void* const* h_instance, const uint8_t* seed16, zk_proof** out);
/* The same with halo2's instance slices as they are: h_instance[i] holds h_instance_len[i] values
 * (not n).  Exactly those values are absorbed into the transcript -- what create_proof and
 * verify_proof do: a circuit with a handful of public inputs absorbs a handful of scalars -- and the
 * column is zero-padded on the device.  More values than usable rows is Error::InstanceTooLarge
 * (status ZK_ERR_INVALID_ARG).  zk_proof_begin / zk_create_proof are the special case "every
 * usable row is a public input".                                                                  */
int zk_proof_begin_instances(zk_ctx* ctx, const zk_pk* pk, const void* const* h_instance, const uint32_t* h_instance_len, const uint8_t* seed16, zk_proof** out);
/* External transcript (halo2's `T: TranscriptWrite<G1Affine, Challenge255>` argument of
 * create_proof): by default the session runs halo2's Blake2bWrite itself and zk_proof_finish
 * returns the proof bytes.  With a vtable every transcript operation is forwarded to the host
 * language's own transcript object instead -- Poseidon for the aggregation layers
 * (aggregator/src/core.rs:25-28), Keccak for EVM proofs (prover/src/common/prover/evm.rs:67) --
 * which then also owns the proof bytes (zk_proof_finish reports length 0).  Points are 64 B
 * affine (x || y, Montgomery limbs, identity = zeros), scalars 32 B Montgomery Fr; a callback
 * returns 0 on success.  Call right after zk_proof_begin: what begin absorbed (vk representative,
 * instance values) is replayed into the external transcript.                                       */
/* Built-in transcripts of the session (the three the reference's call sites instantiate): Blake2b
 * (Blake2bWrite + Challenge255, the default [REF circuit-benchmarks/src/super_circuit.rs:112]),
 * Poseidon (snark-verifier PoseidonTranscript<NativeLoader, _> with POSEIDON_SPEC: gen_snark_shplonk
 * [REF prover/src/common/prover/utils.rs:31], [REF aggregator/src/core.rs:91-92]) and EVM (snark-verifier
 * EvmTranscript, Keccak-256, 64-byte big-endian points: gen_evm_proof_shplonk
 * [REF prover/src/common/prover/evm.rs:67]).  Call right after zk_proof_begin.  A Poseidon / EVM
 * transcript cannot absorb the identity point (as upstream): zk_proof_finish then fails.          */
enum { ZK_TRANSCRIPT_BLAKE2B = 0, ZK_TRANSCRIPT_POSEIDON = 1, ZK_TRANSCRIPT_EVM = 2 };
int zk_proof_set_transcript_kind(zk_ctx* ctx, zk_proof* proof, int kind);
/* The same transcripts as host-only objects (no context, no device): what a caller needs to run
 * halo2's verifier-side transcript logic, or to cross-check its own transcript, without Rust.
 * Points 64 B affine Montgomery, scalars 32 B Montgomery Fr; status 0 = OK.                        */
typedef struct zk_transcript zk_transcript;
zk_transcript* zk_transcript_new(int kind);
void zk_transcript_free(zk_transcript* t);
int zk_transcript_common_point(zk_transcript* t, const void* affine64);
int zk_transcript_common_scalar(zk_transcript* t, const void* fr32);
int zk_transcript_write_point(zk_transcript* t, const void* affine64);
int zk_transcript_write_scalar(zk_transcript* t, const void* fr32);
int zk_transcript_squeeze(zk_transcript* t, void* fr32_out);
/* bytes written so far (the proof); the pointer stays valid until the next write or free          */
size_t zk_transcript_proof(const zk_transcript* t, const void** data);
/* host-only hashes behind them: Keccak-256 and the Poseidon permutation (5 x 32 B Montgomery Fr)   */
int zk_host_keccak256(const void* data, size_t len, void* out32);
int zk_host_poseidon_permute(void* state5_fr32);
/* the same generator at width 3 (R_F = 8, R_P = 57): the permutation under the reference's Poseidon code hash; its value on
 * (0, 0, 0) is POSEIDON_CODE_HASH_EMPTY (eth-types/src/lib.rs:278), which is what pins the constant generation          */
int zk_host_poseidon_permute_width3(void* state3_fr32);
typedef struct zk_transcript_vtable {
    int (*common_point)(void* user, const void* affine64);
    int (*common_scalar)(void* user, const void* fr32);
    int (*write_point)(void* user, const void* affine64);
    int (*write_scalar)(void* user, const void* fr32);
    int (*squeeze_challenge)(void* user, void* fr32_out);
} zk_transcript_vtable;
int zk_proof_set_transcript(zk_ctx* ctx, zk_proof* proof, const zk_transcript_vtable* vtable, void* user);
/* Multi-GPU proving: one process per GPU, every rank runs the SAME session calls on
 * the same key, witness and seed.  Rank r then commits only columns r, r + world, ... and evaluates
 * only cosets r, r + world, ... of the quotient; commitments (64 B each) and finished cosets
 * (2^k x 32 B each) are exchanged through `gather`, so all ranks produce the same transcript and
 * the same proof bytes as an unsharded session.  gather(user, send, bytes, recv) must all-gather
 * `bytes` bytes from every rank into recv (world x bytes, rank-major) and return 0 -- e.g.
 * torch.distributed.all_gather_into_tensor over RCCL, see zkevm-circuits_amd/sharding.py.
 * Call between zk_proof_begin and the first advice phase.                                          */
typedef int (*zk_allgather_fn)(void* user, const void* h_send, size_t bytes, void* h_recv);
int zk_proof_set_sharding(zk_ctx* ctx, zk_proof* proof, uint32_t rank, uint32_t world, zk_allgather_fn gather, void* user);
/* ---- collectives inside the library: RCCL over xGMI, one process per GPU (csrc/comm.hip) ----------
 * For host languages without a collective library of their own (the Rust shim): rank 0 makes a
 * 128-byte unique id (zk_comm_unique_id) and hands it to every rank by any out-of-band channel;
 * zk_comm_init joins the communicator (ncclCommInitRank) on the context's device.  After that
 *   zk_proof_set_sharding_comm   shards a proving session over the communicator: commitments
 *                                all-gathered per transcript round, advice columns and finished
 *                                quotient cosets device to device, everything stream-ordered;
 *   zk_ntt_sharded(.., NULL, NULL) uses one grouped ncclSend / ncclRecv all-to-all;
 *   zk_comm_allgather / _alltoall  are the raw collectives over device buffers.
 * librccl is loaded on first use; a single-GPU deployment never needs it.                          */
int zk_comm_unique_id(void* out128);
int zk_comm_init(zk_ctx* ctx, const void* id128, uint32_t rank, uint32_t world);
int zk_comm_destroy(zk_ctx* ctx);
int zk_comm_allgather(zk_ctx* ctx, const void* d_send, size_t bytes, void* d_recv);
int zk_comm_alltoall(zk_ctx* ctx, const void* d_send, size_t bytes_per_peer, void* d_recv);
int zk_proof_set_sharding_comm(zk_ctx* ctx, zk_proof* proof);
/* Optional, after zk_proof_set_sharding: an all-gather of DEVICE buffers (same signature, device
 * pointers; RCCL over xGMI).  Each rank then uploads only its own 1/world of the advice columns and
 * the ranks exchange them over the fabric, instead of every rank pulling every column over its own
 * PCIe link.  The library drains its stream before each call; the callback must return only when
 * d_recv is complete.                                                                              */
int zk_proof_set_device_gather(zk_ctx* ctx, zk_proof* proof, zk_allgather_fn gather_dev, void* user);
/* commits the advice columns of the current phase (h_cols[j] = advice column col_index[j], exactly
 * the columns of that phase) and writes the challenges that become usable after it to
 * h_challenges (32 B each, Montgomery Fr, challenge-index order).  With h_challenges given,
 * *num_challenges must hold the buffer's capacity (in challenges) on entry -- a phase with more is
 * refused, nothing is written -- and receives the number written (zk_pk_shape: total challenges). */
int zk_proof_advice_phase(zk_ctx* ctx, zk_proof* proof, const uint32_t* col_index, const void* const* h_cols, uint32_t ncols,
                          void* h_challenges, uint32_t* num_challenges);
/* The same phase for witness columns RESIDENT ON THE DEVICE (device pointers, n x 32 B each, Montgomery form): a witness generated
 * on the GPU or uploaded ahead of the proof; small / dense columns are judged on the device.  Same proof bytes.  flags = 0: the
 * session copies the columns device to device into its own buffers (the caller's stay untouched).  ZK_ADVICE_DEV_IN_PLACE: the
 * session works in the caller's buffers -- it overwrites their last blinding_factors + 1 rows (halo2 puts blinding values there; a
 * witness holds nothing in them) and reads them until zk_proof_finish / zk_proof_abort returns; no copy, n x 32 B less device
 * memory per column.
 * Sharded session with a device all-gather (zk_proof_set_sharding_comm, or zk_proof_set_sharding + zk_proof_set_device_gather):
 * a rank needs only the columns it OWNS -- position j of the phase's columns in ascending column index, j % world == rank; those
 * it commits and sends to the other ranks over the fabric.  d_cols[j] of a column the rank does not own may be NULL (col_index
 * still lists every column of the phase on every rank).                                                                           */
#define ZK_ADVICE_DEV_IN_PLACE 1u
int zk_proof_advice_phase_dev(zk_ctx* ctx, zk_proof* proof, const uint32_t* col_index, const void* const* d_cols, uint32_t ncols, uint32_t flags,
                              void* h_challenges, uint32_t* num_challenges);
/* consumes the session (freed on success and on failure)                                         */
int zk_proof_finish(zk_ctx* ctx, zk_proof* proof, void* h_proof, size_t proof_cap, size_t* proof_len);
void zk_proof_abort(zk_ctx* ctx, zk_proof* proof);

/* ---- host-only wire formats and aggregation arithmetic (csrc/wire.hip, csrc/host_pairing.hpp) ------
 * No context, no device.  SerdeFormat codes: 0 Processed, 1 RawBytes, 2 RawBytesUnchecked.          */
/* instances <-> concatenated 32-byte big-endian words [REF prover/src/proof.rs:77-85,126-138]         */
int zk_host_instances_encode(const void* fr_mont, size_t n, void* out_be);
int zk_host_instances_decode(const void* in_be, size_t n, void* fr_mont_out);
/* The prover's `Proof` wire object [REF prover/src/proof.rs:25-35,99-104] -- the body of full_proof_<name>.json:
 * {"proof": base64, "instances": base64 of the 32-byte big-endian words (zk_host_instances_encode), "vk": base64 of
 * VerifyingKey::write(Processed) (zk_host_vk_write), "git_version": string or null}, serde_json's compact form in the struct's
 * field order; base64 as [REF eth-types/src/lib.rs:71-91] (standard alphabet, padded).  write: out may be NULL (size query);
 * git_version NULL = null.  read: accepts compact or pretty JSON with the four keys in any order (git_version may be absent);
 * every output buffer is optional, the *_len arguments hold capacities on entry and lengths on return.  The objects built
 * around it -- ChunkProof / BatchProof with snark-verifier's `protocol` -- stay on the Rust side. */
int zk_host_proof_json_write(const void* proof, size_t proof_len, const void* instances_be, size_t instances_len, const void* vk, size_t vk_len,
                             const char* git_version, char* out, size_t cap, size_t* len);
int zk_host_proof_json_read(const char* json, size_t json_len, void* proof, size_t* proof_len, void* instances_be, size_t* instances_len, void* vk, size_t* vk_len,
                            char* git_version, size_t git_cap, int* has_git_version);
/* Instances as the prover's JSON matrix [REF prover/src/io.rs:28-56]: `serialize_instance` = serde_json (compact) of
 * Vec<Vec<Vec<u8>>> -- per instance column a list of elements, each the 32 little-endian bytes of Fr::to_bytes as numbers.
 * (`load_instances` [REF prover/src/io.rs:128-142] holds a list of such matrices: one more pair of brackets.)  write: out may
 * be NULL (size query).  read: NULL outputs = counts only; values of all columns one after the other, Montgomery Fr. */
int zk_host_instances_json_write(const void* const* cols_fr_mont, const size_t* lens, size_t ncols, char* out, size_t cap, size_t* len);
int zk_host_instances_json_read(const char* json, size_t json_len, size_t* ncols, size_t* lens_out, size_t lens_cap, void* fr_mont_out, size_t fr_cap, size_t* total);
/* G1 points in halo2curves' SerdeFormat: 32 B compressed (Processed) or 64 B Montgomery limbs       */
int zk_host_g1_encode(const void* affine64, size_t n, int format, void* out);
int zk_host_g1_decode(const void* in, size_t n, int format, void* affine64_out);
/* halo2 VerifyingKey::write / read [REF prover/src/io.rs:97-106]: k and the commitment count as u32
 * big-endian, fixed then permutation commitments, then the selector assignments packed 8 rows/byte  */
int zk_host_vk_write(uint32_t k, const void* fixed_commitments, uint32_t num_fixed, const void* perm_commitments, uint32_t num_perm,
                     const uint8_t* selectors_packed, uint32_t num_selectors, int format, void* out, size_t cap, size_t* len);
int zk_host_vk_read(const void* in, size_t in_len, int format, uint32_t num_perm, uint32_t num_selectors, uint32_t* k, uint32_t* num_fixed,
                    void* fixed_commitments, size_t fixed_cap, void* perm_commitments, uint8_t* selectors_packed);
/* BN254 optimal-ate pairing product: prod e(P_i, Q_i) == 1 ?  (G2: 128 B = x.c0, x.c1, y.c0, y.c1)   */
int zk_host_pairing_check(const void* g1_points, const void* g2_points, size_t n, int* ok);
/* aggregation layers between two GPU proofs [REF aggregator/src/core.rs:48-147]: combine the child
 * snarks' KZG accumulators with powers of a Poseidon-transcript challenge (KzgAs::create_proof,
 * no blinding), decide e(lhs, g2) == e(rhs, s_g2), encode the result as 12 limbs of 88 bits          */
int zk_host_accumulate(const void* lhs_in, const void* rhs_in, size_t n, void* lhs_out, void* rhs_out, void* r_out);
int zk_host_accumulator_check(const void* lhs, const void* rhs, const void* g2, const void* s_g2, int* ok);
int zk_host_accumulator_limbs(const void* lhs, const void* rhs, void* out12_fr);

/* ---- G1 element-wise (tests of the group law; halo2curves G1 Add / Double / Mul) --------------- */
/* out[i] = a[i] + b[i], all affine (n x 64 B) */
int zk_g1_affine_add_vec(zk_ctx* ctx, const void* d_a, const void* d_b, void* d_out, size_t n);
/* out[i] = scalars[i] * bases[i], affine out */
int zk_g1_mul_vec(zk_ctx* ctx, const void* d_bases, const void* d_scalars, void* d_out, size_t n);

/* ---- introspection ------------------------------------------------------------------------------ */
const char* zk_version(void);
/* fills name (<= len) with the device name, CU count and HBM bytes of the ctx device */
int zk_device_info(zk_ctx* ctx, char* name, size_t len, int* cu_count, size_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* ZKMI355_H */
