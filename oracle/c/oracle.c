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
