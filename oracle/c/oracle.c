/*
 * CPU oracle (TEST INFRASTRUCTURE ONLY) -- C restatement of the BN254 arithmetic on the Halo2/KZG
 * hot path.  Checker for the HIP path; never linked into or called from the product library.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * The algorithms restated here live in crates that are NOT under /root/reference:
 *   halo2curves 0.1.0 @ a495a7b   src/bn256/{fr,fq,curve}.rs, src/derive/{field,curve}.rs
 *   halo2_proofs 1.1.0 @ e5ddf67  src/arithmetic.rs (best_fft, best_multiexp, eval_polynomial,
 *                                 kate_division), src/poly/domain.rs (EvaluationDomain)
 *   [REF Cargo.lock:2214-2216,2239-2241]; reference call sites: SURVEY 8a A1-A5.
 * Memory layout = halo2curves in-memory layout: field element = 4 x u64 LE limbs, Montgomery
 * form (R = 2^256); G1Affine = {x, y} (identity = (0,0)); G1 = Jacobian {x, y, z}.
 *
 * Parity status: pinned against oracle/bn254.py (tests/test_oracle_c.py), which is itself
 * pinned by the reference's golden vectors G3-G6 (see oracle/bn254.py header).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;
typedef struct { uint64_t m[4]; uint64_t inv; fe one; fe r2; } fld;

static const fld FR = {
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0xc2e1f593efffffffULL,
    {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}},
    {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}},
};
static const fld FQ = {
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0x87d20782e4866389ULL,
    {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}},
    {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}},
};

static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, sizeof(fe)) == 0; }
static inline int ge_mod(const uint64_t *a, const uint64_t *m) {
    for (int i = 3; i >= 0; --i) { if (a[i] > m[i]) return 1; if (a[i] < m[i]) return 0; }
    return 1;
}
static inline void sub_mod_raw(uint64_t *a, const uint64_t *m) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - m[i] - (uint64_t)br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
static inline void fe_add(fe *o, const fe *a, const fe *b, const fld *F) {
    u128 c = 0; uint64_t t[4];
    for (int i = 0; i < 4; ++i) { c += (u128)a->l[i] + b->l[i]; t[i] = (uint64_t)c; c >>= 64; }
    if (ge_mod(t, F->m)) sub_mod_raw(t, F->m);   /* a,b < m < 2^254: no carry out of limb 3 */
    memcpy(o->l, t, 32);
}
static inline void fe_sub(fe *o, const fe *a, const fe *b, const fld *F) {
    u128 br = 0; uint64_t t[4];
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)br; t[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)t[i] + F->m[i]; t[i] = (uint64_t)c; c >>= 64; } }
    memcpy(o->l, t, 32);
}
static inline void fe_neg(fe *o, const fe *a, const fld *F) {
    if (fe_is_zero(a)) { *o = *a; return; }
    fe z = {{0, 0, 0, 0}}; fe_sub(o, &z, a, F);
}
static inline void fe_dbl(fe *o, const fe *a, const fld *F) { fe_add(o, a, a, F); }
/* CIOS Montgomery product (word-for-word the same schedule as oracle/bn254.py:mont_mul_cios) */
static inline void fe_mul(fe *o, const fe *a, const fe *b, const fld *F) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->inv;
        c = (u128)m * F->m[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * F->m[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || ge_mod(t, F->m)) sub_mod_raw(t, F->m);
    memcpy(o->l, t, 32);
}
static inline void fe_sqr(fe *o, const fe *a, const fld *F) { fe_mul(o, a, a, F); }
static void fe_pow(fe *o, const fe *a, const uint64_t e[4], const fld *F) {
    fe r = F->one, b = *a;
    for (int i = 0; i < 256; ++i) {
        if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(&r, &r, &b, F);
        fe_sqr(&b, &b, F);
    }
    *o = r;
}
static void fe_inv(fe *o, const fe *a, const fld *F) {   /* Fermat; inv(0) = 0 */
    uint64_t e[4] = {F->m[0] - 2, F->m[1], F->m[2], F->m[3]};
    fe_pow(o, a, e, F);
}
static void fe_to_canon(uint64_t out[4], const fe *a, const fld *F) {
    fe one = {{1, 0, 0, 0}}, t; fe_mul(&t, a, &one, F); memcpy(out, t.l, 32);
}

/* ------------------------------------------------------------------ exported field helpers */
#define API __attribute__((visibility("default")))
static const fld *pick(int which) { return which ? &FQ : &FR; }

API void orc_fe_mul_vec(int which, const fe *a, const fe *b, fe *o, size_t n) { const fld *F = pick(which); for (size_t i = 0; i < n; ++i) fe_mul(&o[i], &a[i], &b[i], F); }
API void orc_fe_add_vec(int which, const fe *a, const fe *b, fe *o, size_t n) { const fld *F = pick(which); for (size_t i = 0; i < n; ++i) fe_add(&o[i], &a[i], &b[i], F); }
API void orc_fe_sub_vec(int which, const fe *a, const fe *b, fe *o, size_t n) { const fld *F = pick(which); for (size_t i = 0; i < n; ++i) fe_sub(&o[i], &a[i], &b[i], F); }
API void orc_fe_inv_vec(int which, const fe *a, fe *o, size_t n) { const fld *F = pick(which); for (size_t i = 0; i < n; ++i) fe_inv(&o[i], &a[i], F); }
API void orc_fe_to_mont_vec(int which, const fe *a, fe *o, size_t n) { const fld *F = pick(which); for (size_t i = 0; i < n; ++i) fe_mul(&o[i], &a[i], &F->r2, F); }
API void orc_fe_from_mont_vec(int which, const fe *a, fe *o, size_t n) { const fld *F = pick(which); for (size_t i = 0; i < n; ++i) fe_to_canon(o[i].l, &a[i], F); }

/* splitmix64 stream -> field elements (identical to oracle/bn254.py:rand_fr_stream); output is
 * the raw integer v < r interpreted directly as the in-memory (Montgomery) limbs. */
static inline uint64_t splitmix64(uint64_t *st) {
    uint64_t z = (*st += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
API void orc_rand_fr_stream(uint64_t seed, fe *o, size_t n) {
    uint64_t st = seed;
    for (size_t i = 0; i < n; ++i) {
        for (int w = 0; w < 4; ++w) o[i].l[w] = splitmix64(&st);
        o[i].l[3] &= 0x3fffffffffffffffULL;
        if (ge_mod(o[i].l, FR.m)) sub_mod_raw(o[i].l, FR.m);
    }
}

/* ------------------------------------------------------------------ NTT (halo2 best_fft) */
static inline size_t bitrev(size_t x, unsigned bits) { size_t r = 0; for (unsigned i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

/* Restates arithmetic::best_fft: bit-reverse, twiddles[i] = omega^i, radix-2 DIT stages.
 * Natural order in -> natural order out.  Stages are data-parallel over butterflies (the
 * reference splits recursively over Rayon threads; the arithmetic per output is identical). */
API void orc_best_fft(fe *a, const fe *omega, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t k = 0; k < n; ++k) { size_t rk = bitrev(k, log_n); if (k < rk) { fe t = a[k]; a[k] = a[rk]; a[rk] = t; } }
    size_t half_n = n / 2 ? n / 2 : 1;
    fe *tw = (fe *)malloc(sizeof(fe) * half_n);
    /* powers of omega, built in parallel blocks */
    {
        size_t blk = 1024;
        #pragma omp parallel for schedule(static)
        for (size_t b0 = 0; b0 < half_n; b0 += blk) {
            uint64_t e[4] = {b0, 0, 0, 0}; fe w; fe_pow(&w, omega, e, &FR);
            size_t end = b0 + blk < half_n ? b0 + blk : half_n;
            for (size_t i = b0; i < end; ++i) { tw[i] = w; fe_mul(&w, &w, omega, &FR); }
        }
    }
    size_t chunk = 2, tchunk = n / 2;
    for (unsigned s = 0; s < log_n; ++s) {
        size_t half = chunk / 2;
        #pragma omp parallel for schedule(static)
        for (size_t bf = 0; bf < n / 2; ++bf) {
            size_t blk = bf / half, i = bf % half, lo = blk * chunk + i, hi = lo + half;
            fe t; fe_mul(&t, &a[hi], &tw[i * tchunk], &FR);
            fe u = a[lo];
            fe_add(&a[lo], &u, &t, &FR); fe_sub(&a[hi], &u, &t, &FR);
        }
        chunk *= 2; tchunk /= 2;
    }
    free(tw);
}

API void orc_scale_vec(fe *a, const fe *s, size_t n) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) fe_mul(&a[i], &a[i], s, &FR);
}

/* a[i] *= g^i  (distribute_powers; used by coeff_to_extended with g = zeta) */
API void orc_distribute_powers(fe *a, const fe *g, size_t n) {
    size_t blk = 4096;
    #pragma omp parallel for schedule(static)
    for (size_t b0 = 0; b0 < n; b0 += blk) {
        uint64_t e[4] = {b0, 0, 0, 0}; fe w; fe_pow(&w, g, e, &FR);
        size_t end = b0 + blk < n ? b0 + blk : n;
        for (size_t i = b0; i < end; ++i) { fe_mul(&a[i], &a[i], &w, &FR); fe_mul(&w, &w, g, &FR); }
    }
}

/* arithmetic::eval_polynomial (Horner, serial definition) */
API void orc_eval_polynomial(const fe *c, size_t n, const fe *x, fe *out) {
    fe acc = {{0, 0, 0, 0}};
    for (size_t i = n; i-- > 0;) { fe_mul(&acc, &acc, x, &FR); fe_add(&acc, &acc, &c[i], &FR); }
    *out = acc;
}

/* arithmetic::kate_division: q = (f - f(z)) / (X - z), n-1 coefficients */
API void orc_kate_division(const fe *c, size_t n, const fe *z, fe *q) {
    fe tmp = {{0, 0, 0, 0}};
    for (size_t i = n - 1; i-- > 0;) { fe t; fe_mul(&t, &tmp, z, &FR); fe_add(&tmp, &c[i + 1], &t, &FR); q[i] = tmp; }
}

/* ff::BatchInvert semantics: zeros stay zero, everything else inverted */
API void orc_batch_invert(fe *a, size_t n) {
    fe *pre = (fe *)malloc(sizeof(fe) * n); fe acc = FR.one;
    for (size_t i = 0; i < n; ++i) { pre[i] = acc; if (!fe_is_zero(&a[i])) fe_mul(&acc, &acc, &a[i], &FR); }
    fe inv; fe_inv(&inv, &acc, &FR);
    for (size_t i = n; i-- > 0;) { if (fe_is_zero(&a[i])) continue; fe t; fe_mul(&t, &inv, &pre[i], &FR); fe_mul(&inv, &inv, &a[i], &FR); a[i] = t; }
    free(pre);
}

/* inclusive-from-one prefix product as used for the permutation / lookup grand products:
 * z[0] = 1, z[i+1] = z[i] * a[i]  (n outputs from the first n-1 inputs... caller slices) */
API void orc_prefix_product(const fe *a, fe *z, size_t n) {
    fe acc = FR.one;
    for (size_t i = 0; i < n; ++i) { z[i] = acc; fe_mul(&acc, &acc, &a[i], &FR); }
}
API void orc_prefix_sum(const fe *a, fe *z, size_t n) {
    fe acc = {{0, 0, 0, 0}};
    for (size_t i = 0; i < n; ++i) { z[i] = acc; fe_add(&acc, &acc, &a[i], &FR); }
}

/* ------------------------------------------------------------------ vector helpers of the restated CPU prover (oracle/cpu_prover.py)
 * Row-parallel (OpenMP) like halo2's `parallelize` chunks.  All values Montgomery form. */
API void orc_fe_mul_vec_mt(const fe *a, const fe *b, fe *o, size_t n) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) fe_mul(&o[i], &a[i], &b[i], &FR);
}
API void orc_fe_add_vec_mt(const fe *a, const fe *b, fe *o, size_t n) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) fe_add(&o[i], &a[i], &b[i], &FR);
}
API void orc_fe_sub_vec_mt(const fe *a, const fe *b, fe *o, size_t n) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) fe_sub(&o[i], &a[i], &b[i], &FR);
}
/* o = a * s + b   (b may be NULL: o = a * s) */
API void orc_fe_scale_add(const fe *a, const fe *s, const fe *b, fe *o, size_t n) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) { fe t; fe_mul(&t, &a[i], s, &FR); if (b) fe_add(&o[i], &t, &b[i], &FR); else o[i] = t; }
}
API void orc_fe_add_scalar(const fe *a, const fe *s, fe *o, size_t n) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) fe_add(&o[i], &a[i], s, &FR);
}
/* o[i] = a[(i + shift) mod n]  (a rotation of a column on a cyclic domain) */
API void orc_fe_rotate(const fe *a, long long shift, fe *o, size_t n) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) { long long j = ((long long)i + shift) % (long long)n; if (j < 0) j += (long long)n; o[i] = a[j]; }
}
/* Postfix expression over columns, every row (plonk::evaluation restated as a stack machine; the same instruction set as the
 * key blob: 1 PUSH_COL (a = index into cols, b = rotation), 2 PUSH_CONST (a = index into consts), 3 ADD, 4 SUB, 5 MUL, 6 NEG).
 * Row i reads cols[a][(i + rot * stride) mod n].  Returns 0, or -1 on a malformed program. */
API int orc_eval_program(const uint32_t *prog, size_t n_instr, const fe *const *cols, const fe *consts, size_t n, long long stride, fe *out) {
    int depth = 0, maxd = 0;
    for (size_t p = 0; p < n_instr; ++p) {
        uint32_t op = prog[3 * p];
        if (op == 1 || op == 2) { if (++depth > maxd) maxd = depth; }
        else if (op >= 3 && op <= 5) { if (depth < 2) return -1; --depth; }
        else if (op == 6) { if (depth < 1) return -1; }
        else return -1;
    }
    if (depth != 1 || maxd > 64) return -1;
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        fe st[64]; int sp = 0;
        for (size_t p = 0; p < n_instr; ++p) {
            const uint32_t op = prog[3 * p], a = prog[3 * p + 1]; const int32_t rot = (int32_t)prog[3 * p + 2];
            switch (op) {
                case 1: { long long j = ((long long)i + (long long)rot * stride) % (long long)n; if (j < 0) j += (long long)n; st[sp++] = cols[a][j]; break; }
                case 2: st[sp++] = consts[a]; break;
                case 3: fe_add(&st[sp - 2], &st[sp - 2], &st[sp - 1], &FR); --sp; break;
                case 4: fe_sub(&st[sp - 2], &st[sp - 2], &st[sp - 1], &FR); --sp; break;
                case 5: fe_mul(&st[sp - 2], &st[sp - 2], &st[sp - 1], &FR); --sp; break;
                default: fe_neg(&st[sp - 1], &st[sp - 1], &FR); break;
            }
        }
        out[i] = st[0];
    }
    return 0;
}

/* ------------------------------------------------------------------ G1 */
typedef struct { fe x, y; } aff;
typedef struct { fe x, y, z; } jac;

static inline int aff_is_id(const aff *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline void jac_set_id(jac *p) { memset(p, 0, sizeof(*p)); p->y = FQ.one; }  /* (0,1,0) like halo2curves */
static inline int jac_is_id(const jac *p) { return fe_is_zero(&p->z); }

static void jac_double(jac *o, const jac *p) {     /* dbl-2009-l, a = 0 */
    if (jac_is_id(p)) { jac_set_id(o); return; }
    fe a, b, c, d, e, f, t, x3, y3, z3;
    fe_sqr(&a, &p->x, &FQ); fe_sqr(&b, &p->y, &FQ); fe_sqr(&c, &b, &FQ);
    fe_add(&t, &p->x, &b, &FQ); fe_sqr(&t, &t, &FQ); fe_sub(&t, &t, &a, &FQ); fe_sub(&t, &t, &c, &FQ); fe_dbl(&d, &t, &FQ);
    fe_dbl(&e, &a, &FQ); fe_add(&e, &e, &a, &FQ); fe_sqr(&f, &e, &FQ);
    fe_mul(&z3, &p->y, &p->z, &FQ); fe_dbl(&z3, &z3, &FQ);
    fe_dbl(&t, &d, &FQ); fe_sub(&x3, &f, &t, &FQ);
    fe_dbl(&c, &c, &FQ); fe_dbl(&c, &c, &FQ); fe_dbl(&c, &c, &FQ);
    fe_sub(&t, &d, &x3, &FQ); fe_mul(&y3, &e, &t, &FQ); fe_sub(&y3, &y3, &c, &FQ);
    o->x = x3; o->y = y3; o->z = z3;
}
static void jac_add(jac *o, const jac *p, const jac *q) {    /* add-2007-bl */
    if (jac_is_id(p)) { *o = *q; return; }
    if (jac_is_id(q)) { *o = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
    fe_sqr(&z1z1, &p->z, &FQ); fe_sqr(&z2z2, &q->z, &FQ);
    fe_mul(&u1, &p->x, &z2z2, &FQ); fe_mul(&u2, &q->x, &z1z1, &FQ);
    fe_mul(&s1, &p->y, &q->z, &FQ); fe_mul(&s1, &s1, &z2z2, &FQ);
    fe_mul(&s2, &q->y, &p->z, &FQ); fe_mul(&s2, &s2, &z1z1, &FQ);
    if (fe_eq(&u1, &u2)) { if (fe_eq(&s1, &s2)) jac_double(o, p); else jac_set_id(o); return; }
    fe_sub(&h, &u2, &u1, &FQ); fe_dbl(&i, &h, &FQ); fe_sqr(&i, &i, &FQ); fe_mul(&j, &h, &i, &FQ);
    fe_sub(&r, &s2, &s1, &FQ); fe_dbl(&r, &r, &FQ); fe_mul(&v, &u1, &i, &FQ);
    fe_sqr(&x3, &r, &FQ); fe_sub(&x3, &x3, &j, &FQ); fe_sub(&x3, &x3, &v, &FQ); fe_sub(&x3, &x3, &v, &FQ);
    fe_sub(&t, &v, &x3, &FQ); fe_mul(&y3, &r, &t, &FQ); fe_mul(&t, &s1, &j, &FQ); fe_dbl(&t, &t, &FQ); fe_sub(&y3, &y3, &t, &FQ);
    fe_add(&z3, &p->z, &q->z, &FQ); fe_sqr(&z3, &z3, &FQ); fe_sub(&z3, &z3, &z1z1, &FQ); fe_sub(&z3, &z3, &z2z2, &FQ); fe_mul(&z3, &z3, &h, &FQ);
    o->x = x3; o->y = y3; o->z = z3;
}
static void jac_madd(jac *o, const jac *p, const aff *q) {   /* madd-2007-bl */
    if (aff_is_id(q)) { *o = *p; return; }
    if (jac_is_id(p)) { o->x = q->x; o->y = q->y; o->z = FQ.one; return; }
    fe z1z1, u2, s2, h, hh, i, j, r, v, t, x3, y3, z3;
    fe_sqr(&z1z1, &p->z, &FQ); fe_mul(&u2, &q->x, &z1z1, &FQ);
    fe_mul(&s2, &q->y, &p->z, &FQ); fe_mul(&s2, &s2, &z1z1, &FQ);
    if (fe_eq(&p->x, &u2)) { if (fe_eq(&p->y, &s2)) jac_double(o, p); else jac_set_id(o); return; }
    fe_sub(&h, &u2, &p->x, &FQ); fe_sqr(&hh, &h, &FQ); fe_dbl(&i, &hh, &FQ); fe_dbl(&i, &i, &FQ); fe_mul(&j, &h, &i, &FQ);
    fe_sub(&r, &s2, &p->y, &FQ); fe_dbl(&r, &r, &FQ); fe_mul(&v, &p->x, &i, &FQ);
    fe_sqr(&x3, &r, &FQ); fe_sub(&x3, &x3, &j, &FQ); fe_sub(&x3, &x3, &v, &FQ); fe_sub(&x3, &x3, &v, &FQ);
    fe_sub(&t, &v, &x3, &FQ); fe_mul(&y3, &r, &t, &FQ); fe_mul(&t, &p->y, &j, &FQ); fe_dbl(&t, &t, &FQ); fe_sub(&y3, &y3, &t, &FQ);
    fe_add(&z3, &p->z, &h, &FQ); fe_sqr(&z3, &z3, &FQ); fe_sub(&z3, &z3, &z1z1, &FQ); fe_sub(&z3, &z3, &hh, &FQ);
    o->x = x3; o->y = y3; o->z = z3;
}
static void jac_to_aff(aff *o, const jac *p) {
    if (jac_is_id(p)) { memset(o, 0, sizeof(*o)); return; }
    fe zi, zi2, zi3; fe_inv(&zi, &p->z, &FQ); fe_sqr(&zi2, &zi, &FQ); fe_mul(&zi3, &zi2, &zi, &FQ);
    fe_mul(&o->x, &p->x, &zi2, &FQ); fe_mul(&o->y, &p->y, &zi3, &FQ);
}

API void orc_g1_jac_add_vec(const jac *a, const jac *b, jac *o, size_t n) { for (size_t i = 0; i < n; ++i) jac_add(&o[i], &a[i], &b[i]); }
API void orc_g1_jac_madd_vec(const jac *a, const aff *b, jac *o, size_t n) { for (size_t i = 0; i < n; ++i) jac_madd(&o[i], &a[i], &b[i]); }
API void orc_g1_jac_double_vec(const jac *a, jac *o, size_t n) { for (size_t i = 0; i < n; ++i) jac_double(&o[i], &a[i]); }
API void orc_g1_to_affine_vec(const jac *a, aff *o, size_t n) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) jac_to_aff(&o[i], &a[i]);
}
API int orc_g1_is_on_curve(const aff *p) {
    if (aff_is_id(p)) return 1;
    fe y2, x3, b3 = {{3, 0, 0, 0}}; fe_mul(&b3, &b3, &FQ.r2, &FQ);
    fe_sqr(&y2, &p->y, &FQ); fe_sqr(&x3, &p->x, &FQ); fe_mul(&x3, &x3, &p->x, &FQ); fe_add(&x3, &x3, &b3, &FQ);
    return fe_eq(&y2, &x3);
}
/* scalar given in Montgomery form (as halo2 holds it); double-and-add, MSB first */
static void g1_mul_aff(jac *o, const aff *p, const fe *scalar_mont) {
    uint64_t k[4]; fe_to_canon(k, scalar_mont, &FR);
    jac acc; jac_set_id(&acc);
    for (int i = 255; i >= 0; --i) { jac_double(&acc, &acc); if ((k[i >> 6] >> (i & 63)) & 1) jac_madd(&acc, &acc, p); }
    *o = acc;
}
API void orc_g1_mul_vec(const aff *p, const fe *scalars, aff *o, size_t n) {
    #pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < n; ++i) { jac t; g1_mul_aff(&t, &p[i], &scalars[i]); jac_to_aff(&o[i], &t); }
}
/* ParamsKZG::unsafe_setup_with_s: g[i] = s^i * G1gen */
API void orc_srs_powers(const fe *s, aff *g, size_t n) {
    aff gen; fe one = {{1, 0, 0, 0}}, two = {{2, 0, 0, 0}};
    fe_mul(&gen.x, &one, &FQ.r2, &FQ); fe_mul(&gen.y, &two, &FQ.r2, &FQ);
    fe *pw = (fe *)malloc(sizeof(fe) * n); fe cur = FR.one;
    for (size_t i = 0; i < n; ++i) { pw[i] = cur; fe_mul(&cur, &cur, s, &FR); }
    #pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < n; ++i) { jac t; g1_mul_aff(&t, &gen, &pw[i]); jac_to_aff(&g[i], &t); }
    free(pw);
}

/* SURVEY 8(d) config 2, second base set: n pseudo-random affine points with no known discrete logarithms --
 * x_i = splitmix64 stream (seed, index) reduced below p, incremented until x^3 + 3 is a square; y = the root
 * (x^3 + 3)^((p + 1) / 4) (p = 3 mod 4), negated when the stream's next bit says so.  Montgomery form, as every
 * base at the ABI.  Each point depends only on (seed, i): the loop is parallel and reproducible. */
API void orc_hash_to_curve_points(uint64_t seed, aff *o, size_t n) {
    fe three_c = {{3, 0, 0, 0}}, one_c = {{1, 0, 0, 0}}, three, one;
    fe_mul(&three, &three_c, &FQ.r2, &FQ); fe_mul(&one, &one_c, &FQ.r2, &FQ);
    uint64_t e[4];                              /* (p + 1) / 4 */
    { unsigned __int128 c = 1; uint64_t t[4]; for (int i = 0; i < 4; ++i) { c += FQ.m[i]; t[i] = (uint64_t)c; c >>= 64; }
      for (int i = 0; i < 4; ++i) e[i] = (t[i] >> 2) | (i < 3 ? t[i + 1] << 62 : 0); }
    #pragma omp parallel for schedule(dynamic, 256)
    for (size_t i = 0; i < n; ++i) {
        uint64_t st = seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(i + 1));
        fe x, rhs, y, chk;
        for (int w = 0; w < 4; ++w) x.l[w] = splitmix64(&st);
        x.l[3] &= 0x3fffffffffffffffULL;
        while (ge_mod(x.l, FQ.m)) sub_mod_raw(x.l, FQ.m);       /* raw limbs taken as the Montgomery image: still uniform */
        const int flip = (int)(splitmix64(&st) & 1);
        for (;;) {
            fe_sqr(&rhs, &x, &FQ); fe_mul(&rhs, &rhs, &x, &FQ); fe_add(&rhs, &rhs, &three, &FQ);
            fe_pow(&y, &rhs, e, &FQ);
            fe_sqr(&chk, &y, &FQ);
            if (fe_eq(&chk, &rhs) && !fe_is_zero(&y)) break;
            fe_add(&x, &x, &one, &FQ);
        }
        if (flip) fe_neg(&y, &y, &FQ);
        o[i].x = x; o[i].y = y;
    }
}

/* arithmetic::multiexp_serial restated: unsigned c-bit windows, (256/c)+1 segments from the
 * top with c doublings between, buckets folded by running sum.  Scalars in Montgomery form are
 * first taken to canonical bytes (to_repr) exactly as the reference does. */
static size_t get_at(size_t segment, size_t c, const uint64_t k[4]) {
    size_t skip_bits = segment * c, skip_bytes = skip_bits / 8;
    if (skip_bytes >= 32) return 0;
    uint8_t v[8] = {0}; const uint8_t *bytes = (const uint8_t *)k;
    for (size_t i = 0; i < 8 && skip_bytes + i < 32; ++i) v[i] = bytes[skip_bytes + i];
    uint64_t tmp; memcpy(&tmp, v, 8);
    tmp >>= skip_bits - skip_bytes * 8;
    return (size_t)(tmp % ((uint64_t)1 << c));
}
static void multiexp_serial(const uint64_t (*k)[4], const aff *bases, size_t n, jac *acc) {
    size_t c = n < 4 ? 1 : n < 32 ? 3 : (size_t)ceil(log((double)n));
    size_t segments = 256 / c + 1, nb = ((size_t)1 << c) - 1;
    jac *buckets = (jac *)malloc(sizeof(jac) * nb);
    for (size_t seg = segments; seg-- > 0;) {
        for (size_t i = 0; i < c; ++i) jac_double(acc, acc);
        for (size_t i = 0; i < nb; ++i) jac_set_id(&buckets[i]);
        for (size_t i = 0; i < n; ++i) { size_t d = get_at(seg, c, k[i]); if (d) jac_madd(&buckets[d - 1], &buckets[d - 1], &bases[i]); }
        jac running; jac_set_id(&running);
        for (size_t i = nb; i-- > 0;) { jac_add(&running, &buckets[i], &running); jac_add(acc, acc, &running); }
    }
    free(buckets);
}
/* arithmetic::best_multiexp: split into `threads` contiguous chunks, multiexp_serial each, sum. */
API void orc_best_multiexp(const fe *scalars, const aff *bases, size_t n, int threads, aff *out) {
    uint64_t (*k)[4] = (uint64_t (*)[4])malloc(32 * (n ? n : 1));
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) fe_to_canon(k[i], &scalars[i], &FR);
    if (threads < 1) threads = 1;
    jac total; jac_set_id(&total);
    if (n > (size_t)threads) {
        size_t chunk = n / threads, nchunks = (n + chunk - 1) / chunk;
        jac *res = (jac *)malloc(sizeof(jac) * nchunks);
        #pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
        for (size_t ci = 0; ci < nchunks; ++ci) {
            size_t lo = ci * chunk, hi = lo + chunk < n ? lo + chunk : n;
            jac_set_id(&res[ci]); multiexp_serial(k + lo, bases + lo, hi - lo, &res[ci]);
        }
        for (size_t ci = 0; ci < nchunks; ++ci) jac_add(&total, &total, &res[ci]);
        free(res);
    } else {
        multiexp_serial(k, bases, n, &total);
    }
    jac_to_aff(out, &total);
    free(k);
}
API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
