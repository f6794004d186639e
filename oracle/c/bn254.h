/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * BN254 (alt_bn128) field and G1 arithmetic on the CPU: a plain-C restatement of what
 * the reference reaches through pairing_ce 0.24.2 / ff_ce 0.12.0 (Cargo.lock:1212-1214,
 * 594-596; sources are NOT under /root/reference, so this restates the published
 * algorithms: 4x64-bit Montgomery CIOS, Jacobian short-Weierstrass formulas).
 * Constants are those of SURVEY.md Appendix A.2, re-derived with Python big ints in
 * tests/test_oracle_field.py.
 *
 * In-memory layout (identical to the product's C ABI, include/plonkit_amd.h):
 *   field element = uint64_t[4], little-endian limbs, MONTGOMERY form (R = 2^256)
 *   G1 affine     = x[4] || y[4]; the point at infinity is x = y = 0
 *   G1 jacobian   = X[4] || Y[4] || Z[4]; infinity is Z = 0
 */
#ifndef ORC_BN254_H
#define ORC_BN254_H
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe_t;            /* Fr or Fq, by context */
typedef struct { fe_t x, y; } g1a_t;               /* affine      */
typedef struct { fe_t x, y, z; } g1j_t;            /* jacobian    */

/* ---- generic 4-limb Montgomery field, instantiated for Fr and Fq --------------------- */
#define ORC_DEFINE_FIELD(F, P0, P1, P2, P3, INV, R0, R1, R2_, R3, S0, S1, S2, S3)               \
static const uint64_t F##_P[4]  = {P0, P1, P2, P3};                                             \
static const fe_t F##_ONE = {{R0, R1, R2_, R3}};   /* R mod p   */                              \
static const fe_t F##_RR  = {{S0, S1, S2, S3}};    /* R^2 mod p */                              \
static inline int F##_is_zero(const fe_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; } \
static inline int F##_eq(const fe_t *a, const fe_t *b) { return memcmp(a, b, 32) == 0; }        \
static inline int F##_geq_p(const uint64_t *t) {                                                \
    for (int i = 3; i >= 0; i--) { if (t[i] > F##_P[i]) return 1; if (t[i] < F##_P[i]) return 0; } \
    return 1; }                                                                                 \
static inline void F##_sub_p(uint64_t *t) {                                                     \
    u128 b = 0;                                                                                 \
    for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - F##_P[i] - (uint64_t)b; t[i] = (uint64_t)d; b = (d >> 64) & 1; } } \
static inline void F##_add(fe_t *r, const fe_t *a, const fe_t *b) {                             \
    u128 c = 0; uint64_t t[4];                                                                  \
    for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; t[i] = (uint64_t)c; c >>= 64; } \
    if (F##_geq_p(t)) F##_sub_p(t);                                                             \
    memcpy(r->l, t, 32); }                                                                      \
static inline void F##_sub(fe_t *r, const fe_t *a, const fe_t *b) {                             \
    u128 br = 0; uint64_t t[4];                                                                 \
    for (int i = 0; i < 4; i++) { u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)br; t[i] = (uint64_t)d; br = (d >> 64) & 1; } \
    if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)t[i] + F##_P[i]; t[i] = (uint64_t)c; c >>= 64; } } \
    memcpy(r->l, t, 32); }                                                                      \
static inline void F##_neg(fe_t *r, const fe_t *a) {                                            \
    if (F##_is_zero(a)) { memset(r, 0, 32); return; }                                           \
    u128 br = 0; uint64_t t[4];                                                                 \
    for (int i = 0; i < 4; i++) { u128 d = (u128)F##_P[i] - a->l[i] - (uint64_t)br; t[i] = (uint64_t)d; br = (d >> 64) & 1; } \
    memcpy(r->l, t, 32); }                                                                      \
static inline void F##_dbl(fe_t *r, const fe_t *a) { F##_add(r, a, a); }                        \
/* CIOS Montgomery product: r = a*b*R^-1 mod p, fully reduced */                                \
static inline void F##_mul(fe_t *r, const fe_t *a, const fe_t *b) {                             \
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};                                                         \
    for (int i = 0; i < 4; i++) {                                                               \
        u128 c = 0;                                                                             \
        for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; } \
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);                              \
        uint64_t m = t[0] * (uint64_t)(INV);                                                    \
        c = (u128)m * F##_P[0] + t[0]; c >>= 64;                                                \
        for (int j = 1; j < 4; j++) { c += (u128)m * F##_P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; } \
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);                       \
    }                                                                                           \
    if (t[4] || F##_geq_p(t)) F##_sub_p(t);                                                     \
    memcpy(r->l, t, 32); }                                                                      \
static inline void F##_sqr(fe_t *r, const fe_t *a) { F##_mul(r, a, a); }                        \
static inline void F##_from_canonical(fe_t *r, const fe_t *a) { F##_mul(r, a, &F##_RR); }       \
static inline void F##_to_canonical(fe_t *r, const fe_t *a) {                                   \
    fe_t one = {{1, 0, 0, 0}}; F##_mul(r, a, &one); }                                           \
static inline void F##_from_u64(fe_t *r, uint64_t v) { fe_t t = {{v, 0, 0, 0}}; F##_from_canonical(r, &t); } \
/* r = a^e, e = 4 little-endian limbs (plain integer) */                                        \
static inline void F##_pow(fe_t *r, const fe_t *a, const uint64_t e[4]) {                       \
    fe_t acc = F##_ONE, base = *a;                                                              \
    for (int i = 0; i < 256; i++) {                                                             \
        if ((e[i >> 6] >> (i & 63)) & 1) F##_mul(&acc, &acc, &base);                            \
        F##_sqr(&base, &base); }                                                                \
    *r = acc; }                                                                                 \
/* Fermat inverse; inv(0) = 0 */                                                                \
static inline void F##_inv(fe_t *r, const fe_t *a) {                                            \
    uint64_t e[4] = {F##_P[0] - 2, F##_P[1], F##_P[2], F##_P[3]};                               \
    F##_pow(r, a, e); }

/* Fr: r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001 */
ORC_DEFINE_FIELD(fr,
    0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL,
    0xc2e1f593efffffffULL,
    0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL,
    0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL)

/* Fq: q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47 */
ORC_DEFINE_FIELD(fq,
    0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL,
    0x87d20782e4866389ULL,
    0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL,
    0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL)

/* ---- G1: y^2 = x^3 + 3 over Fq, generator (1, 2)  (contrib/template.sol:9,67-69) ------ */
static inline void g1j_set_inf(g1j_t *p) { memset(p, 0, sizeof *p); p->x = fq_ONE; p->y = fq_ONE; }
static inline int  g1j_is_inf(const g1j_t *p) { return fq_is_zero(&p->z); }
static inline int  g1a_is_inf(const g1a_t *p) { return fq_is_zero(&p->x) && fq_is_zero(&p->y); }
static inline void g1j_from_affine(g1j_t *r, const g1a_t *a) {
    if (g1a_is_inf(a)) { g1j_set_inf(r); return; }
    r->x = a->x; r->y = a->y; r->z = fq_ONE; }

/* dbl-2009-l (a = 0) */
static inline void g1j_double(g1j_t *r, const g1j_t *p) {
    if (g1j_is_inf(p)) { *r = *p; return; }
    fe_t A, B, C, D, E, F, t, x3, y3, z3;
    fq_sqr(&A, &p->x); fq_sqr(&B, &p->y); fq_sqr(&C, &B);
    fq_add(&t, &p->x, &B); fq_sqr(&t, &t); fq_sub(&t, &t, &A); fq_sub(&t, &t, &C); fq_dbl(&D, &t);
    fq_dbl(&E, &A); fq_add(&E, &E, &A);
    fq_sqr(&F, &E);
    fq_dbl(&t, &D); fq_sub(&x3, &F, &t);
    fq_mul(&z3, &p->y, &p->z); fq_dbl(&z3, &z3);
    fq_sub(&t, &D, &x3); fq_mul(&y3, &E, &t);
    fq_dbl(&C, &C); fq_dbl(&C, &C); fq_dbl(&C, &C); fq_sub(&y3, &y3, &C);
    r->x = x3; r->y = y3; r->z = z3; }

/* madd-2007-bl: jacobian += affine */
static inline void g1j_add_mixed(g1j_t *r, const g1j_t *p, const g1a_t *q) {
    if (g1a_is_inf(q)) { *r = *p; return; }
    if (g1j_is_inf(p)) { g1j_from_affine(r, q); return; }
    fe_t Z1Z1, U2, S2, H, HH, I, J, rr, V, t, x3, y3, z3;
    fq_sqr(&Z1Z1, &p->z);
    fq_mul(&U2, &q->x, &Z1Z1);
    fq_mul(&S2, &q->y, &p->z); fq_mul(&S2, &S2, &Z1Z1);
    if (fq_eq(&U2, &p->x)) {
        if (fq_eq(&S2, &p->y)) { g1j_double(r, p); return; }
        g1j_set_inf(r); return; }
    fq_sub(&H, &U2, &p->x);
    fq_sqr(&HH, &H);
    fq_dbl(&I, &HH); fq_dbl(&I, &I);
    fq_mul(&J, &H, &I);
    fq_sub(&rr, &S2, &p->y); fq_dbl(&rr, &rr);
    fq_mul(&V, &p->x, &I);
    fq_sqr(&x3, &rr); fq_sub(&x3, &x3, &J); fq_dbl(&t, &V); fq_sub(&x3, &x3, &t);
    fq_sub(&t, &V, &x3); fq_mul(&y3, &rr, &t); fq_mul(&t, &p->y, &J); fq_dbl(&t, &t); fq_sub(&y3, &y3, &t);
    fq_add(&z3, &p->z, &H); fq_sqr(&z3, &z3); fq_sub(&z3, &z3, &Z1Z1); fq_sub(&z3, &z3, &HH);
    r->x = x3; r->y = y3; r->z = z3; }

/* add-2007-bl: jacobian += jacobian */
static inline void g1j_add(g1j_t *r, const g1j_t *p, const g1j_t *q) {
    if (g1j_is_inf(q)) { *r = *p; return; }
    if (g1j_is_inf(p)) { *r = *q; return; }
    fe_t Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, x3, y3, z3;
    fq_sqr(&Z1Z1, &p->z); fq_sqr(&Z2Z2, &q->z);
    fq_mul(&U1, &p->x, &Z2Z2); fq_mul(&U2, &q->x, &Z1Z1);
    fq_mul(&S1, &p->y, &q->z); fq_mul(&S1, &S1, &Z2Z2);
    fq_mul(&S2, &q->y, &p->z); fq_mul(&S2, &S2, &Z1Z1);
    if (fq_eq(&U1, &U2)) {
        if (fq_eq(&S1, &S2)) { g1j_double(r, p); return; }
        g1j_set_inf(r); return; }
    fq_sub(&H, &U2, &U1);
    fq_dbl(&I, &H); fq_sqr(&I, &I);
    fq_mul(&J, &H, &I);
    fq_sub(&rr, &S2, &S1); fq_dbl(&rr, &rr);
    fq_mul(&V, &U1, &I);
    fq_sqr(&x3, &rr); fq_sub(&x3, &x3, &J); fq_dbl(&t, &V); fq_sub(&x3, &x3, &t);
    fq_sub(&t, &V, &x3); fq_mul(&y3, &rr, &t); fq_mul(&t, &S1, &J); fq_dbl(&t, &t); fq_sub(&y3, &y3, &t);
    fq_add(&z3, &p->z, &q->z); fq_sqr(&z3, &z3); fq_sub(&z3, &z3, &Z1Z1); fq_sub(&z3, &z3, &Z2Z2);
    fq_mul(&z3, &z3, &H);
    r->x = x3; r->y = y3; r->z = z3; }

static inline void g1j_neg(g1j_t *r, const g1j_t *p) { r->x = p->x; r->z = p->z; fq_neg(&r->y, &p->y); }

static inline void g1j_to_affine(g1a_t *r, const g1j_t *p) {
    if (g1j_is_inf(p)) { memset(r, 0, sizeof *r); return; }
    fe_t zi, zi2, zi3;
    fq_inv(&zi, &p->z); fq_sqr(&zi2, &zi); fq_mul(&zi3, &zi2, &zi);
    fq_mul(&r->x, &p->x, &zi2); fq_mul(&r->y, &p->y, &zi3); }

/* k = canonical (non-Montgomery) scalar, 4 LE limbs; plain double-and-add, MSB first */
static inline void g1j_mul_scalar(g1j_t *r, const g1j_t *p, const uint64_t k[4]) {
    g1j_t acc; g1j_set_inf(&acc);
    for (int i = 255; i >= 0; i--) {
        g1j_double(&acc, &acc);
        if ((k[i >> 6] >> (i & 63)) & 1) g1j_add(&acc, &acc, p); }
    *r = acc; }

static inline int g1a_on_curve(const g1a_t *p) {
    if (g1a_is_inf(p)) return 1;
    fe_t y2, x3, b;
    fq_sqr(&y2, &p->y); fq_sqr(&x3, &p->x); fq_mul(&x3, &x3, &p->x);
    fq_from_u64(&b, 3); fq_add(&x3, &x3, &b);
    return fq_eq(&y2, &x3); }

#endif
