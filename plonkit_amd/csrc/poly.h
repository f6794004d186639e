#pragma once
#include "ctx.h"
namespace plk {

// Scalars marked W are handed over as c * 2^261 (host: 32 * c in the external form), the others in the external form E;
// power tables marked W are filled with 32 * base^e (poly.hip explains the domains)
struct PermArgs {
    Fr *num, *den;                      // out: W domain
    const Fr *w[4], *sigma[4];
    Fr beta, gamma, beta_k[4], fix;     // beta: W; gamma, beta_k[j] = beta * k_j: E; fix = E(2^25), i.e. 2^281
    uint32_t n, log_n;
    PowTable tw;                        // omega table, W
};

struct CheckArgs {
    const Fr *values, *q[7];
    const uint32_t *vars[4];
    uint32_t n, num_inputs;
    uint32_t *flag;
};

// The quotient kernel computes on the 9 x 29-bit layer (field29_dev.h), whose product is a*b*2^-261 while vectors in
// HBM carry the factor 2^256.  Instead of converting anything per proof, the constant vectors cached in plk_setup
// are stored pre-scaled (2^261: q_a..q_d, q_dnext, sigma_j, L0, the coset points x; 2^266: q_m, which meets a
// product of two wires) and the host scales the challenges, so that every term lands on 2^256 by itself:
//   alpha_pp = alpha * 2^281 (meets z * four 2^256 factors), alpha2_w = alpha^2 * 2^261, zh_inv_w = 2^261 / Z_H.
constexpr uint32_t QUOTIENT_MAX_DIRECT_PI = 8;
struct QuotientArgs {
    Fr *out;
    const Fr *w[4], *z, *q[7], *sigma[4], *pi, *l0, *x;
    Fr beta, gamma, alpha_pp, alpha2_w, beta_k[4], zh_inv_w[4];
    // beta, alpha_pp and alpha2_w once more as mul_tw3 constants (field29_dev.h: three shifted copies, a product in 108 multiply-adds instead of 162):
    // seven of the kernel's 24 products are by these three per-proof constants (round 6; filled by quotient() from the three fields above)
    uint32_t beta3[27], alpha_pp3[27], alpha2_3[27];
    uint32_t m, log_m;                  // m = 4N
    // public inputs: PI(x) = sum_i in_i * L_i(x) and L_i(x) = L_0(x / omega^i), i.e. on the coset
    // PI[j] = sum_i in_i * L0[j - 4i]: with few inputs the kernel forms it from the cached L0 vector and no
    // PI polynomial is interpolated or extended (pi == nullptr); otherwise pi is its extension.
    uint32_t num_pi;
    Fr pi_in[QUOTIENT_MAX_DIRECT_PI];
    PowTable tw_w;                      // used when x == nullptr: coset points computed on the fly (coset_w = 7 * 2^261)
    Fr coset_w;
};
int32_t scale_const(Fr *out, const Fr *in, const Fr &c_s, uint32_t n, hipStream_t s);            // out_i = in_i * c (one W-layer product), canonical
int32_t coset_points_w(Fr *out, const PowTable &tw_w, uint32_t log_m, const Fr &c_s, uint32_t m, hipStream_t s);   // out_i = c * omega_m^(4 r + k) at the coset-major position i = k * m/4 + r

constexpr uint32_t LINCOMB_MAX = 14;
struct LinCombArgs {
    Fr *out;
    const Fr *p[LINCOMB_MAX];
    Fr s[LINCOMB_MAX];                  // W
    uint32_t unit[LINCOMB_MAX];         // 1: coefficient is one (skip the multiply)
    uint32_t count, n;
    PowTable times_pow;                 // optional (lo != null): out_i = (sum) * base^i — the first step of the division by (x - z)
};

constexpr uint32_t EVAL_MAX = 12;
struct EvalArgs {
    const Fr *poly[EVAL_MAX];
    uint32_t len[EVAL_MAX];
    PowTable pt[EVAL_MAX];              // power table of the evaluation point, W
    uint32_t count, max_blocks;
    Fr *partials;
};

int32_t gather(Fr *out, const Fr *values, const uint32_t *vars, uint32_t n, hipStream_t s);
// out[j][i] = copy[j][i] = values[vars[j][i]] for the four wire columns, one launch
int32_t gather4_dual(Fr *const out[4], Fr *const copy[4], const Fr *values, const uint32_t *const vars[4], uint32_t n, hipStream_t s);
int32_t sigma_from_index(Fr *out, const uint32_t *packed, uint32_t n, uint32_t log_n, const PowTable &tw, const Fr k[4], hipStream_t s);
int32_t check_gates(const CheckArgs &a, hipStream_t s);
// perm.hip: 4 x n packed successors (idx[col * n + row] = col' << 30 | row') of the copy-constraint permutation
int32_t build_permutation_index(plk_ctx *ctx, const uint32_t *const vars[4], uint32_t n, uint64_t num_vars, uint32_t *idx, hipStream_t st);
// the transpiler's temporaries on the device: values[first_tmp + i] = constant_i + sum_k coeff * values[var]  for the
// linear forms recorded at setup (circuit.h: WitnessOp / WitnessTerm, uploaded as they are — 40-byte records)
int32_t eval_witness_ops(Fr *values, const void *ops_dev, const void *terms_dev, uint32_t n_ops, uint32_t first_tmp, hipStream_t s);
// the same when temporary i may read temporary i - 1 (the partial-sum chains of long linear combinations): one lane per run of such
// temporaries, run_start = the index of every run's first temporary (ascending)
int32_t eval_witness_runs(Fr *values, const void *ops_dev, const void *terms_dev, const void *run_start_dev, uint32_t n_runs, uint32_t n_ops, uint32_t first_tmp, hipStream_t s);
int32_t perm_terms(const PermArgs &a, hipStream_t s);
int32_t mul3(Fr *out, const Fr *a, const Fr *b, const Fr &sc, uint32_t n, hipStream_t s);
// out may alias in.  mult: product scan, else sum; reverse: suffix; exclusive: shifted by one
// totals: where the block totals live (default: the context's poly_tmp; two scans in flight on two streams need two)
int32_t scan(plk_ctx *ctx, Fr *out, const Fr *in, uint32_t n, bool mult, bool reverse, bool exclusive, hipStream_t s, DevBuf *totals = nullptr);
// two product scans of equal length in one launch per phase (the grand product's numerator prefix / denominator suffix)
// pre0 / pre1 given: the third phase (block prefixes folded in) is left to mul3_blocks; see poly.hip
constexpr uint32_t POLY_SCAN_BLOCK = 2048;   // elements per block of the scans: a scan of n elements has ceil(n / 2048) prefixes, then its grand total
int32_t scan_pair_mult(plk_ctx *ctx, Fr *out0, const Fr *in0, bool reverse0, bool exclusive0, Fr *out1, const Fr *in1, bool reverse1, bool exclusive1,
                       uint32_t n, hipStream_t s, const Fr **pre0 = nullptr, const Fr **pre1 = nullptr);
int32_t mul3_blocks(Fr *out, const Fr *a, const Fr *b, const Fr *pre_a, const Fr *pre_b, const Fr &sc, uint32_t n, hipStream_t s);
int32_t quotient(const QuotientArgs &a, hipStream_t s);
// data = the four per-coset coefficient vectors u_k of icoset4cm_dev (u_k at data + k*n) -> the 4n coefficients, natural order, in place;
// constants in the W domain: i^-1 (i = omega_4) and s_c = 7^(-N c) / 4
int32_t icoset_combine(Fr *data, uint32_t n, const Fr &iinv_w, const Fr s_w[4], hipStream_t s);
int32_t lincomb(const LinCombArgs &a, hipStream_t s);
int32_t mul_powers(Fr *out, const Fr *in, const PowTable &t, uint32_t shift, uint32_t n, hipStream_t s);
int32_t div_finish(Fr *q, const Fr *suffix, const PowTable &zinv, uint32_t n, hipStream_t s);
int32_t eval_batch(plk_ctx *ctx, EvalArgs a, Fr *results_dev, hipStream_t s);
// fills a caller-provided 2*POW_TAB table with powers of `base`
int32_t fill_pow_table_into(plk_ctx *ctx, const Fr &base, Fr *buf, PowTable *out, hipStream_t s);
// W-domain tables: entries 32 * base^e in the external form
int32_t fill_pow_tables4_into(plk_ctx *ctx, const Fr bases[4], Fr *const bufs[4], PowTable out[4], hipStream_t s);

}  // namespace plk
