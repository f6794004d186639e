/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see bn254.h).  CPU restatement of the arithmetic the
 * reference's prove path reaches in bellman_ce 0.3.2 (git matter-labs/bellman @ 5809cc16,
 * Cargo.lock:109-111 — source NOT under /root/reference).  Call sites in the reference:
 *   src/plonk.rs:104      setup()                 -> orc_fr_ntt (11 iNTT)
 *   src/plonk.rs:123      make_verification_key() -> orc_g1_msm (11 commitments)
 *   src/plonk.rs:140-169  prove / prove_by_steps  -> orc_fr_ntt, orc_g1_msm, vector ops below
 *   src/plonk.rs:41,47    Crs::crs_42             -> orc_crs42
 *   src/plonk.rs:179-185  Crs::from_powers        -> orc_g1_intt
 * The algorithms restated here are the published ones:
 *   best_fft  : serial radix-2 DIT (bit-reverse, log n sweeps) / the 2^log_cpus-way split
 *   dense_multiexp: Pippenger, c = 3 if n < 32 else ceil(ln n), per-thread bucket sets,
 *                   windows processed one after another, c doublings between windows,
 *                   zero scalars skipped, scalar == 1 added directly in the first window.
 * Parity is pinned by tests/test_oracle_golden.py (vk.bin / proof.bin byte-for-byte).
 */
#include "bn254.h"
#include <stdlib.h>
#include <math.h>
#include <omp.h>

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ scalar helpers ---- */
EXPORT void orc_fr_mul(fe_t *r, const fe_t *a, const fe_t *b) { fr_mul(r, a, b); }
EXPORT void orc_fr_add(fe_t *r, const fe_t *a, const fe_t *b) { fr_add(r, a, b); }
EXPORT void orc_fr_sub(fe_t *r, const fe_t *a, const fe_t *b) { fr_sub(r, a, b); }
EXPORT void orc_fr_inv(fe_t *r, const fe_t *a) { fr_inv(r, a); }
EXPORT void orc_fq_mul(fe_t *r, const fe_t *a, const fe_t *b) { fq_mul(r, a, b); }
EXPORT void orc_fq_add(fe_t *r, const fe_t *a, const fe_t *b) { fq_add(r, a, b); }
EXPORT void orc_fq_sub(fe_t *r, const fe_t *a, const fe_t *b) { fq_sub(r, a, b); }
EXPORT void orc_fq_inv(fe_t *r, const fe_t *a) { fq_inv(r, a); }

EXPORT void orc_fr_from_canonical(fe_t *out, const fe_t *in, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) fr_from_canonical(&out[i], &in[i]); }
EXPORT void orc_fr_to_canonical(fe_t *out, const fe_t *in, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) fr_to_canonical(&out[i], &in[i]); }
EXPORT void orc_fq_from_canonical(fe_t *out, const fe_t *in, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) fq_from_canonical(&out[i], &in[i]); }
EXPORT void orc_fq_to_canonical(fe_t *out, const fe_t *in, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) fq_to_canonical(&out[i], &in[i]); }

/* ------------------------------------------------------------------------- domain ----- */
/* omega(2^k) = (7^((r-1)/2^28))^(2^(28-k))   (2-adicity 28, generator 7) */
static const fe_t FR_ROOT28_CANON = {{0xd34f1ed960c37c9cULL, 0x3215cf6dd39329c8ULL,
                                      0x98865ea93dd31f74ULL, 0x03ddb9f5166d18b7ULL}};
EXPORT void orc_fr_omega(fe_t *out, uint32_t log_n) {
    fe_t w; fr_from_canonical(&w, &FR_ROOT28_CANON);
    for (uint32_t i = log_n; i < 28; i++) fr_sqr(&w, &w);
    *out = w; }

static void fr_pow_u64(fe_t *r, const fe_t *a, uint64_t e) { uint64_t ee[4] = {e, 0, 0, 0}; fr_pow(r, a, ee); }

/* ------------------------------------------------------------------------- NTT -------- */
static inline uint32_t bitrev(uint32_t x, uint32_t bits) {
    uint32_t r = 0; for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

static void serial_ntt(fe_t *a, const fe_t *omega, uint32_t log_n) {
    uint32_t n = 1u << log_n;
    for (uint32_t k = 0; k < n; k++) { uint32_t rk = bitrev(k, log_n); if (k < rk) { fe_t t = a[k]; a[k] = a[rk]; a[rk] = t; } }
    uint32_t m = 1;
    for (uint32_t s = 0; s < log_n; s++) {
        fe_t w_m; fr_pow_u64(&w_m, omega, n / (2 * m));
        for (uint32_t k = 0; k < n; k += 2 * m) {
            fe_t w = fr_ONE;
            for (uint32_t j = 0; j < m; j++) {
                fe_t t; fr_mul(&t, &a[k + j + m], &w);
                fr_sub(&a[k + j + m], &a[k + j], &t);
                fr_add(&a[k + j], &a[k + j], &t);
                fr_mul(&w, &w, &w_m); } }
        m *= 2; } }

static void parallel_ntt(fe_t *a, const fe_t *omega, uint32_t log_n, uint32_t log_cpus) {
    uint32_t num_cpus = 1u << log_cpus, log_new_n = log_n - log_cpus, new_n = 1u << log_new_n, n = 1u << log_n;
    fe_t *tmp = calloc((size_t)n, sizeof(fe_t));
    fe_t new_omega; fr_pow_u64(&new_omega, omega, num_cpus);
    #pragma omp parallel for num_threads(num_cpus) schedule(static, 1)
    for (uint32_t j = 0; j < num_cpus; j++) {
        fe_t *t = tmp + (size_t)j * new_n;
        fe_t omega_j, omega_step; fr_pow_u64(&omega_j, omega, j); fr_pow_u64(&omega_step, omega, (uint64_t)j << log_new_n);
        fe_t elt = fr_ONE;
        for (uint32_t i = 0; i < new_n; i++) {
            for (uint32_t s = 0; s < num_cpus; s++) {
                uint32_t idx = (i + (s << log_new_n)) & (n - 1);
                fe_t v; fr_mul(&v, &a[idx], &elt);
                fr_add(&t[i], &t[i], &v);
                fr_mul(&elt, &elt, &omega_step); }
            fr_mul(&elt, &elt, &omega_j); }
        serial_ntt(t, &new_omega, log_new_n); }
    uint32_t mask = num_cpus - 1;
    #pragma omp parallel for num_threads(num_cpus)
    for (uint32_t idx = 0; idx < n; idx++) a[idx] = tmp[(size_t)(idx & mask) * new_n + (idx >> log_cpus)];
    free(tmp); }

static uint32_t log2_floor(uint32_t x) { uint32_t l = 0; while ((1u << (l + 1)) <= x) l++; return l; }

static void best_ntt(fe_t *a, const fe_t *omega, uint32_t log_n, int threads) {
    uint32_t log_cpus = log2_floor(threads < 1 ? 1 : (uint32_t)threads);
    if (log_n <= log_cpus || log_cpus == 0) serial_ntt(a, omega, log_n);
    else parallel_ntt(a, omega, log_n, log_cpus); }

/* a[i] *= g^i */
static void distribute_powers(fe_t *a, uint64_t n, const fe_t *g, int threads) {
    #pragma omp parallel num_threads(threads)
    {
        int t = omp_get_thread_num(), nt = omp_get_num_threads();
        uint64_t chunk = (n + nt - 1) / nt, lo = (uint64_t)t * chunk, hi = lo + chunk > n ? n : lo + chunk;
        if (lo < hi) {
            fe_t x; fr_pow_u64(&x, g, lo);
            for (uint64_t i = lo; i < hi; i++) { fr_mul(&a[i], &a[i], &x); fr_mul(&x, &x, g); } }
    } }

/* Natural order in and out.  inverse=0: a <- evaluations on coset*<omega>; inverse=1: a <- coefficients.
 * coset (Montgomery Fr) may be NULL (= 1).  Mirrors Polynomial::{fft,ifft,coset_fft,icoset_fft}. */
EXPORT void orc_fr_ntt(fe_t *a, uint32_t log_n, int inverse, const fe_t *coset, int threads) {
    uint64_t n = 1ull << log_n;
    fe_t omega; orc_fr_omega(&omega, log_n);
    if (!inverse) {
        if (coset) distribute_powers(a, n, coset, threads);
        best_ntt(a, &omega, log_n, threads);
    } else {
        fe_t omega_inv, n_fe, n_inv; fr_inv(&omega_inv, &omega);
        best_ntt(a, &omega_inv, log_n, threads);
        fr_from_u64(&n_fe, n); fr_inv(&n_inv, &n_fe);
        #pragma omp parallel for num_threads(threads)
        for (uint64_t i = 0; i < n; i++) fr_mul(&a[i], &a[i], &n_inv);
        if (coset) { fe_t gi; fr_inv(&gi, coset); distribute_powers(a, n, &gi, threads); }
    } }

/* O(n^2) definition, for cross-checking the fast path at small n */
EXPORT void orc_fr_dft_naive(fe_t *out, const fe_t *in, uint32_t log_n) {
    uint64_t n = 1ull << log_n; fe_t omega; orc_fr_omega(&omega, log_n);
    for (uint64_t k = 0; k < n; k++) {
        fe_t wk, x = fr_ONE, acc = {{0, 0, 0, 0}}; fr_pow_u64(&wk, &omega, k);
        for (uint64_t i = 0; i < n; i++) { fe_t t; fr_mul(&t, &in[i], &x); fr_add(&acc, &acc, &t); fr_mul(&x, &x, &wk); }
        out[k] = acc; } }

/* ------------------------------------------------------------------- vector ops ------- */
EXPORT void orc_fr_vec_mul(fe_t *r, const fe_t *a, const fe_t *b, uint64_t n, int threads) {
    #pragma omp parallel for num_threads(threads)
    for (uint64_t i = 0; i < n; i++) fr_mul(&r[i], &a[i], &b[i]); }
EXPORT void orc_fr_vec_add(fe_t *r, const fe_t *a, const fe_t *b, uint64_t n, int threads) {
    #pragma omp parallel for num_threads(threads)
    for (uint64_t i = 0; i < n; i++) fr_add(&r[i], &a[i], &b[i]); }
EXPORT void orc_fr_vec_sub(fe_t *r, const fe_t *a, const fe_t *b, uint64_t n, int threads) {
    #pragma omp parallel for num_threads(threads)
    for (uint64_t i = 0; i < n; i++) fr_sub(&r[i], &a[i], &b[i]); }
EXPORT void orc_fr_vec_scale(fe_t *r, const fe_t *a, const fe_t *s, uint64_t n, int threads) {
    #pragma omp parallel for num_threads(threads)
    for (uint64_t i = 0; i < n; i++) fr_mul(&r[i], &a[i], s); }
EXPORT void orc_fr_vec_add_scalar(fe_t *r, const fe_t *a, const fe_t *s, uint64_t n, int threads) {
    #pragma omp parallel for num_threads(threads)
    for (uint64_t i = 0; i < n; i++) fr_add(&r[i], &a[i], s); }
/* r[i] = a[i] + s * b[i] */
EXPORT void orc_fr_vec_axpy(fe_t *r, const fe_t *a, const fe_t *s, const fe_t *b, uint64_t n, int threads) {
    #pragma omp parallel for num_threads(threads)
    for (uint64_t i = 0; i < n; i++) { fe_t t; fr_mul(&t, &b[i], s); fr_add(&r[i], &a[i], &t); } }
/* r[i] = g^i * s */
EXPORT void orc_fr_vec_powers(fe_t *r, const fe_t *g, const fe_t *s, uint64_t n) {
    fe_t x = *s; for (uint64_t i = 0; i < n; i++) { r[i] = x; fr_mul(&x, &x, g); } }
/* Montgomery-trick batch inversion; zeros stay zero */
EXPORT void orc_fr_vec_batch_inv(fe_t *a, uint64_t n) {
    fe_t *pre = malloc(n * sizeof(fe_t)); fe_t acc = fr_ONE;
    for (uint64_t i = 0; i < n; i++) { pre[i] = acc; if (!fr_is_zero(&a[i])) fr_mul(&acc, &acc, &a[i]); }
    fr_inv(&acc, &acc);
    for (uint64_t i = n; i-- > 0;) {
        if (fr_is_zero(&a[i])) continue;
        fe_t t; fr_mul(&t, &acc, &pre[i]); fr_mul(&acc, &acc, &a[i]); a[i] = t; }
    free(pre); }
/* out[0] = 1, out[i+1] = out[i] * a[i]   (n outputs from the first n-1 inputs) */
EXPORT void orc_fr_vec_shifted_prefix_product(fe_t *out, const fe_t *a, uint64_t n) {
    fe_t acc = fr_ONE; for (uint64_t i = 0; i < n; i++) { out[i] = acc; if (i + 1 < n) fr_mul(&acc, &acc, &a[i]); } }
/* Horner evaluation of coefficients at x */
EXPORT void orc_fr_poly_eval(fe_t *out, const fe_t *c, uint64_t n, const fe_t *x) {
    fe_t acc = {{0, 0, 0, 0}};
    for (uint64_t i = n; i-- > 0;) { fr_mul(&acc, &acc, x); fr_add(&acc, &acc, &c[i]); }
    *out = acc; }
/* q(x) = (p(x) - p(z)) / (x - z), synthetic division; q has n-1 coeffs, q[n-1] = 0 */
EXPORT void orc_fr_poly_div_linear(fe_t *q, const fe_t *p, uint64_t n, const fe_t *z) {
    fe_t carry = {{0, 0, 0, 0}};
    for (uint64_t i = n; i-- > 0;) { fe_t t = carry; fr_mul(&carry, &carry, z); fr_add(&carry, &carry, &p[i]); q[i] = t; }
    /* invariant: before step i, carry = q[i]; after it carry = p[i] + z*q[i] = q[i-1]; the last carry is p(z) */
}

/* ------------------------------------------------------------------------- G1 --------- */
EXPORT void orc_g1_generator(g1a_t *out) { fq_from_u64(&out->x, 1); fq_from_u64(&out->y, 2); }
EXPORT int  orc_g1_on_curve(const g1a_t *p) { return g1a_on_curve(p); }
EXPORT void orc_g1_add_affine(g1a_t *r, const g1a_t *a, const g1a_t *b) {
    g1j_t j; g1j_from_affine(&j, a); g1j_add_mixed(&j, &j, b); g1j_to_affine(r, &j); }
EXPORT void orc_g1_neg_affine(g1a_t *r, const g1a_t *a) { r->x = a->x; if (g1a_is_inf(a)) r->y = a->y; else fq_neg(&r->y, &a->y); }
/* k: Montgomery Fr scalar */
EXPORT void orc_g1_mul_affine(g1a_t *r, const g1a_t *a, const fe_t *k_mont) {
    fe_t k; fr_to_canonical(&k, k_mont);
    g1j_t j, o; g1j_from_affine(&j, a); g1j_mul_scalar(&o, &j, k.l); g1j_to_affine(r, &o); }
EXPORT void orc_g1_jac_to_affine(g1a_t *out, const g1j_t *in, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) g1j_to_affine(&out[i], &in[i]); }
EXPORT void orc_g1_jac_add(g1j_t *r, const g1j_t *a, const g1j_t *b) { g1j_add(r, a, b); }

/* batch normalisation with one inversion */
static void batch_to_affine(g1a_t *out, const g1j_t *in, uint64_t n) {
    fe_t *pre = malloc(n * sizeof(fe_t)); fe_t acc = fq_ONE;
    for (uint64_t i = 0; i < n; i++) { pre[i] = acc; if (!g1j_is_inf(&in[i])) fq_mul(&acc, &acc, &in[i].z); }
    fq_inv(&acc, &acc);
    for (uint64_t i = n; i-- > 0;) {
        if (g1j_is_inf(&in[i])) { memset(&out[i], 0, sizeof(g1a_t)); continue; }
        fe_t zi, zi2, zi3; fq_mul(&zi, &acc, &pre[i]); fq_mul(&acc, &acc, &in[i].z);
        fq_sqr(&zi2, &zi); fq_mul(&zi3, &zi2, &zi);
        fq_mul(&out[i].x, &in[i].x, &zi2); fq_mul(&out[i].y, &in[i].y, &zi3); }
    free(pre); }

/* Crs::crs_42 (src/plonk.rs:41,47): out[i] = 42^i * G  — SURVEY.md A.1 [derived] */
EXPORT void orc_crs42(g1a_t *out, uint64_t n, int threads) {
    g1a_t G; orc_g1_generator(&G);
    #pragma omp parallel num_threads(threads)
    {
        int t = omp_get_thread_num(), nt = omp_get_num_threads();
        uint64_t chunk = (n + nt - 1) / nt, lo = (uint64_t)t * chunk, hi = lo + chunk > n ? n : lo + chunk;
        if (lo < hi) {
            g1j_t *buf = malloc((hi - lo) * sizeof(g1j_t));
            fe_t tau, e, ec; fr_from_u64(&tau, 42); fr_pow_u64(&e, &tau, lo); fr_to_canonical(&ec, &e);
            g1j_t g, p; g1j_from_affine(&g, &G); g1j_mul_scalar(&p, &g, ec.l);
            for (uint64_t i = lo; i < hi; i++) {
                buf[i - lo] = p;
                g1j_t p2, p8, p32, s; g1j_double(&p2, &p); g1j_double(&p8, &p2); g1j_double(&p8, &p8);
                g1j_double(&p32, &p8); g1j_double(&p32, &p32);
                g1j_add(&s, &p32, &p8); g1j_add(&p, &s, &p2); }
            batch_to_affine(out + lo, buf, hi - lo);
            free(buf); }
    } }

/* plain definition: sum of double-and-add products (tiny n only) */
EXPORT void orc_g1_msm_naive(g1j_t *out, const g1a_t *bases, const fe_t *scalars_mont, uint64_t n) {
    g1j_t acc; g1j_set_inf(&acc);
    for (uint64_t i = 0; i < n; i++) {
        fe_t k; fr_to_canonical(&k, &scalars_mont[i]);
        g1j_t b, p; g1j_from_affine(&b, &bases[i]); g1j_mul_scalar(&p, &b, k.l); g1j_add(&acc, &acc, &p); }
    *out = acc; }

static inline uint64_t window_bits(const uint64_t k[4], uint32_t skip, uint32_t c) {
    uint32_t limb = skip >> 6, off = skip & 63;
    if (limb >= 4) return 0;
    uint64_t v = k[limb] >> off;
    if (off + c > 64 && limb + 1 < 4) v |= k[limb + 1] << (64 - off);
    return v & ((1ull << c) - 1); }

/* dense_multiexp restatement.  scalars are Montgomery Fr and are converted out first,
 * as commit_using_monomials does before calling multiexp.  Result is Jacobian. */
EXPORT void orc_g1_msm(g1j_t *out, const g1a_t *bases, const fe_t *scalars_mont, uint64_t n, int threads) {
    if (threads < 1) threads = 1;
    fe_t *k = malloc((n ? n : 1) * sizeof(fe_t));
    #pragma omp parallel for num_threads(threads)
    for (uint64_t i = 0; i < n; i++) fr_to_canonical(&k[i], &scalars_mont[i]);
    uint32_t c = n < 32 ? 3 : (uint32_t)ceil(log((double)n));
    uint32_t nwin = (254 + c - 1) / c;
    g1j_t *region = malloc(nwin * sizeof(g1j_t));
    size_t nb = ((size_t)1 << c) - 1;
    uint64_t chunk = (n + threads - 1) / threads; if (chunk == 0) chunk = 1;
    g1j_t *allb = malloc((size_t)threads * nb * sizeof(g1j_t));
    for (uint32_t w = 0; w < nwin; w++) {
        uint32_t skip = w * c; int trivial = (w == 0);
        g1j_t total; g1j_set_inf(&total);
        #pragma omp parallel num_threads(threads)
        {
            int t = omp_get_thread_num();
            uint64_t lo = (uint64_t)t * chunk, hi = lo + chunk > n ? n : lo + chunk;
            g1j_t *b = allb + (size_t)t * nb; g1j_t acc; g1j_set_inf(&acc);
            if (lo < hi) {
                for (size_t i = 0; i < nb; i++) g1j_set_inf(&b[i]);
                for (uint64_t i = lo; i < hi; i++) {
                    const uint64_t *e = k[i].l;
                    if ((e[0] | e[1] | e[2] | e[3]) == 0) continue;
                    if (e[0] == 1 && (e[1] | e[2] | e[3]) == 0) { if (trivial) g1j_add_mixed(&acc, &acc, &bases[i]); continue; }
                    uint64_t d = window_bits(e, skip, c);
                    if (d) g1j_add_mixed(&b[d - 1], &b[d - 1], &bases[i]); }
                g1j_t run; g1j_set_inf(&run);
                for (size_t i = nb; i-- > 0;) { g1j_add(&run, &run, &b[i]); g1j_add(&acc, &acc, &run); }
            }
            #pragma omp critical
            g1j_add(&total, &total, &acc);
        }
        region[w] = total; }
    g1j_t res = region[nwin - 1];
    for (uint32_t w = nwin - 1; w-- > 0;) { for (uint32_t i = 0; i < c; i++) g1j_double(&res, &res); g1j_add(&res, &res, &region[w]); }
    *out = res;
    free(allb); free(region); free(k); }

/* The same sum with the work split by (window, chunk) instead of by chunk alone — what later bellman revisions do
 * (multiexp.rs: one task per window region, each over a slice of the bases).  dense_multiexp above gives EVERY thread its own
 * 2^c - 1 buckets per window, so each thread pays 2 * (2^c - 1) full additions per window whatever its share of the terms:
 * with c = 14 at 2^20 terms that reduction outweighs the useful additions from ~32 threads up, and on a 256-thread host the
 * 0.3.2 shape is slower than on 16 threads.  Here all windows run at once: `groups` = max(1, threads / windows) chunks per
 * window, one task per (window, chunk) with buckets allocated and first touched by the thread that runs the task.  Same c,
 * same zero / one handling, same group element.  This is the baseline a many-core host deserves; kind stays "port". */
EXPORT void orc_g1_msm_wc(g1j_t *out, const g1a_t *bases, const fe_t *scalars_mont, uint64_t n, int threads) {
    if (threads < 1) threads = 1;
    fe_t *k = malloc((n ? n : 1) * sizeof(fe_t));
    #pragma omp parallel for num_threads(threads)
    for (uint64_t i = 0; i < n; i++) fr_to_canonical(&k[i], &scalars_mont[i]);
    uint32_t c = n < 32 ? 3 : (uint32_t)ceil(log((double)n));
    uint32_t nwin = (254 + c - 1) / c;
    size_t nb = ((size_t)1 << c) - 1;
    uint32_t groups = (uint32_t)threads / nwin; if (groups < 1) groups = 1;
    if ((uint64_t)groups > n / 256 + 1) groups = (uint32_t)(n / 256 + 1);
    uint64_t chunk = (n + groups - 1) / groups; if (chunk == 0) chunk = 1;
    uint32_t ntasks = nwin * groups;
    g1j_t *part = malloc((size_t)ntasks * sizeof(g1j_t));
    #pragma omp parallel num_threads(threads)
    {
        g1j_t *b = NULL;
        #pragma omp for schedule(dynamic, 1)
        for (uint32_t task = 0; task < ntasks; task++) {
            if (!b) b = aligned_alloc(64, (nb * sizeof(g1j_t) + 63) & ~(size_t)63);     /* first touched by the thread that owns it */
            uint32_t w = task / groups, g = task % groups;
            uint32_t skip = w * c; int trivial = (w == 0);
            uint64_t lo = (uint64_t)g * chunk, hi = lo + chunk > n ? n : lo + chunk;
            g1j_t acc; g1j_set_inf(&acc);
            if (lo < hi && b) {
                for (size_t i = 0; i < nb; i++) g1j_set_inf(&b[i]);
                for (uint64_t i = lo; i < hi; i++) {
                    const uint64_t *e = k[i].l;
                    if ((e[0] | e[1] | e[2] | e[3]) == 0) continue;
                    if (e[0] == 1 && (e[1] | e[2] | e[3]) == 0) { if (trivial) g1j_add_mixed(&acc, &acc, &bases[i]); continue; }
                    uint64_t d = window_bits(e, skip, c);
                    if (d) g1j_add_mixed(&b[d - 1], &b[d - 1], &bases[i]); }
                g1j_t run; g1j_set_inf(&run);
                for (size_t i = nb; i-- > 0;) { g1j_add(&run, &run, &b[i]); g1j_add(&acc, &acc, &run); }
            }
            part[task] = acc;
        }
        free(b);
    }
    g1j_t res; g1j_set_inf(&res);
    for (uint32_t w = nwin; w-- > 0;) {
        for (uint32_t i = 0; i < c; i++) g1j_double(&res, &res);
        for (uint32_t g = 0; g < groups; g++) g1j_add(&res, &res, &part[(size_t)w * groups + g]); }
    *out = res;
    free(part); free(k); }

/* Crs::<Lagrange>::from_powers (src/plonk.rs:179-185): inverse NTT over G1,
 * out[i] = L_i(tau)*G given in[j] = tau^j*G.  Serial radix-2 DIT on Jacobian points. */
EXPORT void orc_g1_intt(g1a_t *out, const g1a_t *in, uint32_t log_n, int threads) {
    uint32_t n = 1u << log_n;
    g1j_t *a = malloc((size_t)n * sizeof(g1j_t));
    for (uint32_t k = 0; k < n; k++) g1j_from_affine(&a[bitrev(k, log_n)], &in[k]);
    fe_t omega, omega_inv; orc_fr_omega(&omega, log_n); fr_inv(&omega_inv, &omega);
    uint32_t m = 1;
    for (uint32_t s = 0; s < log_n; s++) {
        fe_t w_m; fr_pow_u64(&w_m, &omega_inv, n / (2 * m));
        fe_t *tw = malloc(m * sizeof(fe_t)); fe_t w = fr_ONE;
        for (uint32_t j = 0; j < m; j++) { fr_to_canonical(&tw[j], &w); fr_mul(&w, &w, &w_m); }
        #pragma omp parallel for num_threads(threads) collapse(2)
        for (uint32_t k = 0; k < n; k += 2 * m)
            for (uint32_t j = 0; j < m; j++) {
                g1j_t t, nt, lo = a[k + j];
                if (j == 0) t = a[k + j + m]; else g1j_mul_scalar(&t, &a[k + j + m], tw[j].l);
                g1j_neg(&nt, &t);
                g1j_add(&a[k + j + m], &lo, &nt);
                g1j_add(&a[k + j], &lo, &t); }
        free(tw); m *= 2; }
    fe_t n_fe, n_inv, n_inv_c; fr_from_u64(&n_fe, n); fr_inv(&n_inv, &n_fe); fr_to_canonical(&n_inv_c, &n_inv);
    #pragma omp parallel for num_threads(threads)
    for (uint32_t i = 0; i < n; i++) g1j_mul_scalar(&a[i], &a[i], n_inv_c.l);
    batch_to_affine(out, a, n);
    free(a); }

/* --------------------------------------------------------- circuit front end (CPU baseline) ----
 * The part of `plonkit prove` that runs before the first FFT: read the iden3 .r1cs / .wtns files (src/r1cs_file.rs:100-154,
 * src/reader.rs:124-175), synthesise the circuit gate by gate with the witness (CircomCircuit::synthesize fed to bellman's
 * Width4 adaptor, src/circom_circuit.rs:114-131; rules of SURVEY.md A.3, the same ones oracle/plonk_oracle.py::transpile
 * restates in Python and tests/ compare this code with) and check every gate (is_satisfied_using_one_shot_check,
 * src/plonk.rs:137).  The reference does all of it single-threaded in compiled code; the Python restatement made the CPU
 * baseline of bench.py half interpreter time, so the baseline leg uses these. */
static int fr_from_le32(fe_t *out, const uint8_t *p) {
    fe_t c; memcpy(c.l, p, 32);
    if (fr_geq_p(c.l)) return 0;
    fr_from_canonical(out, &c); return 1; }
static uint32_t rd_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd_u64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* pass 1 (off == NULL): header[0..4] = n_wires, n_pub_out, n_pub_in, n_constraints, n_terms.  pass 2: fills off (3 * nc + 1),
 * wires, coeffs (Montgomery).  Returns 0, or a negative code for a malformed file. */
EXPORT int orc_r1cs_parse(const uint8_t *d, uint64_t len, uint64_t *header, uint64_t *off, uint32_t *wires, fe_t *coeffs) {
    if (len < 12 || memcmp(d, "r1cs", 4) != 0 || rd_u32(d + 4) != 1) return -1;
    uint32_t nsec = rd_u32(d + 8);
    uint64_t o = 12, s_off[4] = {0, 0, 0, 0}, s_len[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < nsec; i++) {
        if (o + 12 > len) return -2;
        uint32_t t = rd_u32(d + o); uint64_t sz = rd_u64(d + o + 4); o += 12;
        if (sz > len - o) return -2;
        if (t >= 1 && t <= 3) { s_off[t] = o; s_len[t] = sz; }
        o += sz; }
    if (!s_off[1] || !s_off[2] || s_len[1] != 64 || rd_u32(d + s_off[1]) != 32) return -3;
    const uint8_t *h = d + s_off[1] + 36;
    uint64_t n_wires = rd_u32(h), n_pub_out = rd_u32(h + 4), n_pub_in = rd_u32(h + 8), nc = rd_u32(h + 24);
    uint64_t p = s_off[2], end = s_off[2] + s_len[2], nt = 0;
    if (off) off[0] = 0;
    for (uint64_t i = 0; i < 3 * nc; i++) {
        if (p + 4 > end) return -4;
        uint32_t k = rd_u32(d + p); p += 4;
        if ((uint64_t)k * 36 > end - p) return -4;
        if (off) for (uint32_t j = 0; j < k; j++) {
            uint32_t w = rd_u32(d + p + 36 * (uint64_t)j);
            if (w >= n_wires) return -5;
            wires[nt + j] = w;
            if (!fr_from_le32(&coeffs[nt + j], d + p + 36 * (uint64_t)j + 4)) return -6; }
        nt += k; p += 36 * (uint64_t)k;
        if (off) off[i + 1] = nt; }
    header[0] = n_wires; header[1] = n_pub_out; header[2] = n_pub_in; header[3] = nc; header[4] = nt;
    return 0; }

/* .wtns -> Montgomery Fr values; out == NULL only reports the count.  Returns the count or a negative code. */
EXPORT int64_t orc_wtns_parse(const uint8_t *d, uint64_t len, fe_t *out, uint64_t cap) {
    if (len < 76 || memcmp(d, "wtns", 4) != 0 || rd_u32(d + 4) > 2 || rd_u32(d + 8) != 2) return -1;
    if (rd_u32(d + 12) != 1 || rd_u64(d + 16) != 40 || rd_u32(d + 24) != 32) return -2;
    uint64_t n = rd_u32(d + 60);
    if (rd_u32(d + 64) != 2 || rd_u64(d + 68) != n * 32 || len < 76 + n * 32) return -3;
    if (!out) return (int64_t)n;
    if (cap < n) return -4;
    for (uint64_t i = 0; i < n; i++) if (!fr_from_le32(&out[i], d + 76 + 32 * i)) return -5;
    return (int64_t)n; }

typedef struct { uint32_t var; fe_t coeff; } term_t;
typedef struct {
    uint32_t *vars; fe_t *q; uint64_t cap, rows;          /* vars[j * cap + row], q[k * cap + row] (q may be NULL) */
    fe_t *values; uint64_t cap_vals, num_vars; int have_values, failed;
} synth_t;
static void sy_gate(synth_t *S, const uint32_t v[4], const fe_t q[7]) {
    if (S->rows >= S->cap) { S->failed = 2; return; }
    for (int j = 0; j < 4; j++) S->vars[(uint64_t)j * S->cap + S->rows] = v[j];
    if (S->q) for (int k = 0; k < 7; k++) S->q[(uint64_t)k * S->cap + S->rows] = q[k];
    S->rows++; }
static uint32_t sy_alloc(synth_t *S, const fe_t *v) {
    if (S->num_vars >= S->cap_vals) { S->failed = 2; return 0; }
    if (S->have_values) S->values[S->num_vars] = *v;
    return (uint32_t)S->num_vars++; }
static fe_t sy_val(const synth_t *S, uint32_t v) { fe_t z = {{0, 0, 0, 0}}; return (S->have_values && v) ? S->values[v] : z; }
/* stable de-duplication; wire 0 (ONE) goes to the constant; zero coefficients dropped */
static size_t sy_split(const uint32_t *w, const fe_t *c, size_t n, fe_t *constant, term_t *out) {
    fe_t z = {{0, 0, 0, 0}}; *constant = z; size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        if (w[i] == 0) { fr_add(constant, constant, &c[i]); continue; }
        size_t j = 0;
        for (; j < m; j++) if (out[j].var == w[i]) { fr_add(&out[j].coeff, &out[j].coeff, &c[i]); break; }
        if (j == m) { out[m].var = w[i]; out[m].coeff = c[i]; m++; } }
    size_t k = 0;
    for (size_t j = 0; j < m; j++) if (!fr_is_zero(&out[j].coeff)) out[k++] = out[j];
    return k; }
static fe_t sy_eval(const synth_t *S, const term_t *lc, size_t n, const fe_t *free_) {
    fe_t s = *free_;
    if (S->have_values) for (size_t i = 0; i < n; i++) { fe_t v = sy_val(S, lc[i].var), t; fr_mul(&t, &lc[i].coeff, &v); fr_add(&s, &s, &t); }
    return s; }
/* lc has room for one more term.  [recollection of bellman's enforce_lc_as_gates; single gate pinned by SURVEY.md A.3] */
static void sy_lc_as_gates(synth_t *S, term_t *lc, size_t n, fe_t free_, int collapse, uint32_t *var_out, fe_t *coeff_out) {
    fe_t zero = {{0, 0, 0, 0}}, minus_one; fr_neg(&minus_one, &fr_ONE);
    if (n == 1 && fr_is_zero(&free_) && collapse) { *var_out = lc[0].var; *coeff_out = lc[0].coeff; return; }
    uint32_t fin = 0;
    if (collapse) { fe_t v = sy_eval(S, lc, n, &free_); fin = sy_alloc(S, &v); lc[n].var = fin; lc[n].coeff = minus_one; n++; }
    uint32_t v[4] = {0, 0, 0, 0}; fe_t q[7];
    for (int i = 0; i < 7; i++) q[i] = zero;
    if (n <= 4) {
        for (size_t i = 0; i < n; i++) { v[i] = lc[i].var; q[i] = lc[i].coeff; }
        q[5] = free_; sy_gate(S, v, q);
    } else {                                              /* UNPINNED: chain through d / d_next */
        size_t pos = 0;
        fe_t s = sy_eval(S, lc, 4, &free_);
        for (int i = 0; i < 4; i++, pos++) { v[i] = lc[pos].var; q[i] = lc[pos].coeff; }
        q[5] = free_; q[6] = minus_one;
        uint32_t nxt = sy_alloc(S, &s);
        sy_gate(S, v, q);
        while (n - pos > 3) {
            for (int i = 0; i < 7; i++) q[i] = zero;
            fe_t prev = sy_val(S, nxt);
            s = sy_eval(S, lc + pos, 3, &prev);
            for (int i = 0; i < 3; i++, pos++) { v[i] = lc[pos].var; q[i] = lc[pos].coeff; }
            v[3] = nxt; q[3] = fr_ONE; q[6] = minus_one;
            uint32_t nn = sy_alloc(S, &s);
            sy_gate(S, v, q);
            nxt = nn; }
        for (int i = 0; i < 7; i++) q[i] = zero;
        v[0] = v[1] = v[2] = 0;
        for (int i = 0; pos < n; i++, pos++) { v[i] = lc[pos].var; q[i] = lc[pos].coeff; }
        v[3] = nxt; q[3] = fr_ONE;
        sy_gate(S, v, q); }
    *var_out = fin; *coeff_out = fr_ONE; }

/* Transpiles constraints [0, nc) into gate rows first_row.. of vars / q (q may be NULL) and, with a witness (Montgomery values
 * indexed by wire; id 0 is the dummy variable = 0), the value of every variable including the temporaries it allocates
 * (ids from num_variables up).  Returns the number of gates, -1 for an unsatisfiable constant constraint, -2 when a
 * capacity is exceeded.  *num_vars_out = num_variables + temporaries. */
EXPORT int64_t orc_transpile(const uint64_t *off, const uint32_t *wires, const fe_t *coeffs, uint64_t nc, uint64_t num_variables,
                             const fe_t *witness, uint32_t *vars_out, fe_t *q_out, uint64_t cap, uint64_t first_row,
                             fe_t *values_out, uint64_t cap_vals, uint64_t *num_vars_out) {
    synth_t S = {vars_out, q_out, cap, first_row, values_out, cap_vals, num_variables, witness != NULL, 0};
    fe_t zero = {{0, 0, 0, 0}};
    if (witness) {
        if (cap_vals < num_variables) return -2;
        memcpy(values_out, witness, num_variables * sizeof(fe_t));
        values_out[0] = zero; }
    size_t max_terms = 1;
    for (uint64_t i = 0; i < 3 * nc; i++) if (off[i + 1] - off[i] > max_terms) max_terms = off[i + 1] - off[i];
    term_t *al = malloc((max_terms + 1) * sizeof(term_t)), *bl = malloc((max_terms + 1) * sizeof(term_t)),
           *cl = malloc((2 * max_terms + 1) * sizeof(term_t));
    uint32_t *mw = malloc(2 * max_terms * sizeof(uint32_t)); fe_t *mc = malloc(2 * max_terms * sizeof(fe_t));
    int64_t rc = 0;
    for (uint64_t idx = 0; idx < nc && !S.failed; idx++) {
        const uint64_t a0 = off[3 * idx], b0 = off[3 * idx + 1], c0 = off[3 * idx + 2], c1 = off[3 * idx + 3];
        if ((b0 == a0 || c0 == b0) && c1 == c0) continue;                        /* src/circom_circuit.rs:121-122 */
        fe_t ac, bc, cc;
        size_t na = sy_split(wires + a0, coeffs + a0, b0 - a0, &ac, al);
        size_t nb = sy_split(wires + b0, coeffs + b0, c0 - b0, &bc, bl);
        size_t ncl = sy_split(wires + c0, coeffs + c0, c1 - c0, &cc, cl);
        uint32_t dv; fe_t dc, q[7];
        for (int i = 0; i < 7; i++) q[i] = zero;
        if (na == 0 && nb == 0) {
            fe_t t, fr_; fr_mul(&t, &ac, &bc); fr_sub(&fr_, &cc, &t);
            if (ncl == 0) { if (!fr_is_zero(&fr_)) { rc = -1; break; } }
            else sy_lc_as_gates(&S, cl, ncl, fr_, 0, &dv, &dc);
        } else if (na == 0 || nb == 0) {                                         /* UNPINNED: constant * LC = LC */
            const fe_t *kk = na == 0 ? &ac : &bc; const term_t *lin = na == 0 ? bl : al; size_t nl = na == 0 ? nb : na;
            const fe_t *lin_c = na == 0 ? &bc : &ac;
            size_t m = 0;
            for (size_t i = 0; i < nl; i++, m++) { mw[m] = lin[i].var; fr_mul(&mc[m], &lin[i].coeff, kk); }
            for (size_t i = 0; i < ncl; i++, m++) { mw[m] = cl[i].var; fr_neg(&mc[m], &cl[i].coeff); }
            fe_t t, fr_, dummy; fr_mul(&t, kk, lin_c); fr_sub(&fr_, &t, &cc);
            size_t nm = sy_split(mw, mc, m, &dummy, cl);
            if (nm) sy_lc_as_gates(&S, cl, nm, fr_, 0, &dv, &dc);
            else if (!fr_is_zero(&fr_)) { rc = -1; break; }
        } else {
            int same = na == 1 && nb == 1 && al[0].var == bl[0].var && (ncl == 0 || (ncl == 1 && cl[0].var == al[0].var));
            if (same) {                                                          /* UNPINNED: quadratic gate */
                fe_t a1 = al[0].coeff, b1 = bl[0].coeff, c1f = ncl ? cl[0].coeff : zero, t1, t2;
                uint32_t v[4] = {al[0].var, al[0].var, 0, 0};
                fr_mul(&t1, &ac, &b1); fr_mul(&t2, &a1, &bc); fr_add(&t1, &t1, &t2); fr_sub(&q[0], &t1, &c1f);
                fr_mul(&q[4], &a1, &b1);
                fr_mul(&t1, &ac, &bc); fr_sub(&q[5], &t1, &cc);
                sy_gate(&S, v, q);
            } else {
                uint32_t av, bv, cv; fe_t acoef, bcoef, ccoef;
                sy_lc_as_gates(&S, al, na, ac, 1, &av, &acoef);
                sy_lc_as_gates(&S, bl, nb, bc, 1, &bv, &bcoef);
                fr_mul(&q[4], &acoef, &bcoef);
                if (ncl == 0) { uint32_t v[4] = {av, bv, 0, 0}; fr_neg(&q[5], &cc); sy_gate(&S, v, q); }
                else { sy_lc_as_gates(&S, cl, ncl, cc, 1, &cv, &ccoef); uint32_t v[4] = {av, bv, cv, 0}; fr_neg(&q[2], &ccoef); sy_gate(&S, v, q); }
            }
        }
    }
    free(al); free(bl); free(cl); free(mw); free(mc);
    if (rc < 0) return rc;
    if (S.failed) return -2;
    *num_vars_out = S.num_vars;
    return (int64_t)(S.rows - first_row); }

/* cols[j][r] = values[vars[j][r]] for r < n (vars with leading dimension cap) */
EXPORT void orc_gather_columns(fe_t *cols, const uint32_t *vars, uint64_t cap, const fe_t *values, uint64_t n, int threads) {
    #pragma omp parallel for num_threads(threads) collapse(2)
    for (int j = 0; j < 4; j++) for (uint64_t r = 0; r < n; r++) cols[(uint64_t)j * n + r] = values[vars[(uint64_t)j * cap + r]]; }

/* is_satisfied_using_one_shot_check: every row r < n satisfies
 * q_a a + q_b b + q_c c + q_d d + q_m a b + q_const + q_d_next d(r + 1) + (r < num_inputs ? a : 0) = 0.
 * sel = 7 selector VALUE vectors of length n; cols = 4 wire-value vectors of length n.  Returns 1 / 0. */
EXPORT int orc_check_gates(const fe_t *cols, const fe_t *sel, uint64_t n, uint64_t num_inputs, int threads) {
    int ok = 1;
    #pragma omp parallel for num_threads(threads) reduction(&&: ok)
    for (uint64_t r = 0; r < n; r++) {
        fe_t acc = sel[5 * n + r], t, ab;
        for (int j = 0; j < 4; j++) { fr_mul(&t, &sel[(uint64_t)j * n + r], &cols[(uint64_t)j * n + r]); fr_add(&acc, &acc, &t); }
        fr_mul(&ab, &cols[r], &cols[n + r]); fr_mul(&t, &sel[4 * n + r], &ab); fr_add(&acc, &acc, &t);
        if (r + 1 < n) { fr_mul(&t, &sel[6 * n + r], &cols[3 * n + r + 1]); fr_add(&acc, &acc, &t); }
        if (r < num_inputs) fr_add(&acc, &acc, &cols[r]);
        ok = ok && fr_is_zero(&acc); }
    return ok; }

/* ----------------------------------------------------------------------- keccak ------- */
static const uint64_t KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
static inline uint64_t rol(uint64_t x, int s) { return s ? (x << s) | (x >> (64 - s)) : x; }
static void keccak_f(uint64_t st[25]) {
    for (int round = 0; round < 24; round++) {
        uint64_t C[5], D[5], B[25];
        for (int x = 0; x < 5; x++) C[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
        for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rol(C[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) st[i] ^= D[i % 5];
        for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++)
            B[y + 5 * ((2 * x + 3 * y) % 5)] = rol(st[x + 5 * y], KROT[x + 5 * y]);
        for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++)
            st[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
        st[0] ^= KRC[round]; } }
/* Ethereum keccak-256 (pad 0x01), as contrib/template.sol:285-303 uses it */
EXPORT void orc_keccak256(uint8_t out[32], const uint8_t *in, uint64_t len) {
    uint64_t st[25]; memset(st, 0, sizeof st); const uint64_t rate = 136;
    uint8_t block[136];
    while (len >= rate) { for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, in + 8 * i, 8); st[i] ^= w; } keccak_f(st); in += rate; len -= rate; }
    memset(block, 0, rate); memcpy(block, in, len); block[len] ^= 0x01; block[rate - 1] ^= 0x80;
    for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, block + 8 * i, 8); st[i] ^= w; } keccak_f(st);
    memcpy(out, st, 32); }
