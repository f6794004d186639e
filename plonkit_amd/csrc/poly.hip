// Element-wise / scan / reduction kernels over Fr vectors for the device-resident prover rounds.
// They replace the Polynomial<Fr, _> helpers bellman_ce's prover calls between FFTs and multiexps
// (grand product, batch inversion, quotient assembly, Horner evaluation, linear combinations,
// division by (x - z)); reference call site: prove_by_steps, src/plonk.rs:152-159.  Formulas are
// those of SURVEY.md Appendix A.4.  All of them are HBM-streaming passes (32 B per operand per
// point) with a handful of modular multiplies per point.
#include "ctx.h"
#include "poly.h"
#include "field29_dev.h"
#include <utility>
#include <type_traits>

namespace plk {

constexpr int PT = 256;                 // threads per block
constexpr int EPT = 8;                  // elements per thread in scans / reductions
constexpr int BLOCK_ELEMS = PT * EPT;
static_assert(BLOCK_ELEMS == (int)POLY_SCAN_BLOCK, "poly.h: POLY_SCAN_BLOCK");

// Compile-time loop: f(integral_constant<int, 0>) .. f(integral_constant<int, N - 1>).  `#pragma unroll` is only a request: the loops over
// the four wire columns in k_perm_terms and k_quotient (two products of the 29-bit layer per trip) were left rolled, and a rolled loop that
// indexes register arrays (w[j], bkx[j]) sends them to scratch memory — 304 B per lane in k_quotient until round 4.

__device__ __forceinline__ Fr pow2l_(const PowTable &t, uint32_t e) {
    return mul(load_fp(t.lo + (e & (POW_TAB - 1))), load_fp(t.hi + (e >> POW_SPLIT)));
}
__device__ __forceinline__ FrW9 ldw(const Fr *p) { return unpack<FrW>(load_fp(p)); }
__device__ __forceinline__ FrW9 cw(const Fr &c) { return unpack<FrW>(c); }
// base^e from a table filled in the W domain: W(lo) * W(hi) * 2^-261 = W(lo * hi)
__device__ __forceinline__ FrW9 pow2l_w_(const PowTable &t, uint32_t e) {
    return mulw(ldw(t.lo + (e & (POW_TAB - 1))), ldw(t.hi + (e >> POW_SPLIT)));
}
__device__ __forceinline__ void stw(Fr *p, const FrW9 &v) { store_fp(p, pack<FrParams>(v)); }     // v normalised, < 2^256

// ------------------------------------------------------------------ gather / permutation
__global__ void __launch_bounds__(PT) k_gather(Fr *out, const Fr *values, const uint32_t *vars, uint32_t n) {
    uint32_t i = blockIdx.x * PT + threadIdx.x;
    if (i < n) store_fp(out + i, load_fp(values + vars[i]));
}

// the four wire columns of a proof in ONE launch, each written twice (values for the grand product, a copy that the iNTT turns
// into coefficients in place): eight launches of ~12 us each were pure launch latency at the head of round 1
struct Gather4 { Fr *out[4]; Fr *copy[4]; const uint32_t *vars[4]; };
__global__ void __launch_bounds__(PT) k_gather4(Gather4 a, const Fr *values, uint32_t n) {
    const uint32_t i = blockIdx.x * PT + threadIdx.x, j = blockIdx.y;
    if (i >= n) return;
    const Fr v = load_fp(values + a.vars[j][i]);
    store_fp(a.out[j] + i, v);
    store_fp(a.copy[j] + i, v);
}

// sigma_j(omega^i) = k_col * omega^row with (col,row) packed as col<<30 | row
__global__ void __launch_bounds__(PT) k_sigma_from_index(Fr *out, const uint32_t *packed, uint32_t n, uint32_t log_n, PowTable tw, Fr k0, Fr k1, Fr k2, Fr k3) {
    uint32_t i = blockIdx.x * PT + threadIdx.x;
    if (i >= n) return;
    uint32_t p = packed[i], col = p >> 30, row = p & 0x3fffffffu;
    Fr w = pow2l_(tw, row << (MAX_LOG_N - log_n));
    Fr k = col == 0 ? k0 : (col == 1 ? k1 : (col == 2 ? k2 : k3));
    store_fp(out + i, col == 0 ? w : mul(w, k));
}

// num_i = prod_j (w_j + beta*k_j*omega^i + gamma) ; den_i = prod_j (w_j + beta*sigma_j + gamma), both written in the W
// domain (the product scans that follow are closed there).  a.tw = omega table in the W domain, a.beta = W(beta), a.beta_k[j]
// and a.gamma in E; a.fix = 2^281: three E x E products leave 2^241, the fourth by 2^281 lands on 2^261.
// 17 products of the 29-bit layer per row (the 8 x 32-bit version: 15 products of 1.8x the cost).
__global__ void __launch_bounds__(PT) k_perm_terms(PermArgs a) {
    uint32_t i = blockIdx.x * PT + threadIdx.x;
    if (i >= a.n) return;
    const FrW9 wi = pow2l_w_(a.tw, i << (MAX_LOG_N - a.log_n));
    const FrW9 gamma = cw(a.gamma), beta = cw(a.beta), fix = cw(a.fix);
    FrW9 num, den;
    static_for<4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const FrW9 wg = addw(ldw(a.w[j] + i), gamma);                                   // raw sums: limbs < 3 * 2^29 feed mulw's left side
        const FrW9 fn = addw(wg, mulw(wi, cw(a.beta_k[j])));
        const FrW9 fd = addw(wg, mulw(ldw(a.sigma[j] + i), beta));
        if (j == 0) { num = fn; den = fd; }
        else { num = mulw(num, normw(fn)); den = mulw(den, normw(fd)); }
    });
    stw(a.num + i, csub_p(mulw(num, fix)));
    stw(a.den + i, csub_p(mulw(den, fix)));
}

// z_i = A_i * C_i * s:  A, C in the W domain (scan outputs), s = E(1 / total)  ->  E, canonical
__global__ void __launch_bounds__(PT) k_mul3(Fr *out, const Fr *a, const Fr *b, Fr s, uint32_t n) {
    uint32_t i = blockIdx.x * PT + threadIdx.x;
    if (i < n) stw(out + i, csub_p(mulw(mulw(ldw(a + i), ldw(b + i)), cw(s))));
}

// the same with the third phase of the two product scans folded in (scan_pair_mult with pre0 / pre1): a = block-local PREFIX scan in memory order,
// b = block-local scan of the REVERSED vector, pre_a / pre_b their block prefixes in scan order — one pass over memory and one launch less in round 2
__global__ void __launch_bounds__(PT) k_mul3_blocks(Fr *out, const Fr *a, const Fr *b, const Fr *pre_a, const Fr *pre_b, Fr s, uint32_t n) {
    uint32_t i = blockIdx.x * PT + threadIdx.x;
    if (i >= n) return;
    const FrW9 pa = ldw(pre_a + i / BLOCK_ELEMS), pb = ldw(pre_b + (n - 1 - i) / BLOCK_ELEMS);
    stw(out + i, csub_p(mulw(mulw(mulw(ldw(a + i), pa), mulw(ldw(b + i), pb)), cw(s))));
}

// -------------------------------------------------------------------------------- scans
// product scans: elements in the W domain (closed under mulw), kept lazily reduced (< 1.01 p < 2^256) in registers, LDS and
// the intermediate arrays, canonical in the final store of phase 3 / phase 1; sums: the 8 x 32-bit modular addition
template <bool MULT> __device__ __forceinline__ Fr op(const Fr &a, const Fr &b) {
    if (MULT) return pack<FrParams>(mulw(unpack<FrW>(a), unpack<FrW>(b)));
    return add(a, b);
}
template <bool MULT> __device__ __forceinline__ Fr fin(const Fr &a) { return MULT ? pack<FrParams>(csub_p(unpack<FrW>(a))) : a; }
template <bool MULT> __device__ __forceinline__ Fr ident() { return MULT ? pack<FrParams>(w_one<FrW>()) : Fr::zero(); }   // W(1) for products

// phase 1: per-block scan of BLOCK_ELEMS elements; writes the block-local (inclusive or exclusive)
// result and the block total.  reverse: logical index i maps to memory n-1-i (suffix scans).
template <bool MULT>
__device__ __forceinline__ void scan_local_body(Fr *out, const Fr *in, Fr *block_tot, uint32_t n, int reverse, int exclusive) {
    __shared__ __attribute__((aligned(16))) Fr sh[PT];
    const uint32_t tid = threadIdx.x, base = blockIdx.x * BLOCK_ELEMS + tid * EPT;
    Fr v[EPT];
    Fr run = ident<MULT>();
    static_for<EPT>([&](auto K) {                               // (a rolled loop would keep v[] in scratch memory)
        constexpr int k = decltype(K)::value;
        uint32_t i = base + k;
        Fr x = i < n ? load_fp(in + (reverse ? n - 1 - i : i)) : ident<MULT>();
        if (exclusive) { v[k] = run; run = op<MULT>(run, x); }
        else { run = op<MULT>(run, x); v[k] = run; }
    });
    sh[tid] = run;
    __syncthreads();
    Fr incl = run;
    for (int off = 1; off < PT; off <<= 1) {
        Fr o = (int)tid >= off ? sh[tid - off] : ident<MULT>();
        __syncthreads();
        if ((int)tid >= off) { incl = op<MULT>(o, incl); sh[tid] = incl; }
        __syncthreads();
    }
    Fr excl = tid ? sh[tid - 1] : ident<MULT>();
    if (tid == PT - 1) store_fp(block_tot + blockIdx.x, incl);
    static_for<EPT>([&](auto K) {
        constexpr int k = decltype(K)::value;
        uint32_t i = base + k;
        if (i < n) store_fp(out + (reverse ? n - 1 - i : i), fin<MULT>(op<MULT>(excl, v[k])));
    });
}

template <bool MULT>
__global__ void __launch_bounds__(PT) k_scan_local(Fr *out, const Fr *in, Fr *block_tot, uint32_t n, int reverse, int exclusive) {
    scan_local_body<MULT>(out, in, block_tot, n, reverse, exclusive);
}
// TWO scans of equal length in one launch per phase (blockIdx.y picks the scan): the kernels are chains of dependent products
// run by two waves per SIMD — latency, not throughput — so the numerator's prefix scan and the denominator's suffix scan of the
// grand product cost the time of one (round 4: 0.52 -> 0.27 ms of round 2).
struct ScanPair { Fr *out[2]; const Fr *in[2]; Fr *tot[2]; int reverse[2], exclusive[2]; };
template <bool MULT>
__global__ void __launch_bounds__(PT) k_scan_local_pair(ScanPair a, uint32_t n) {
    const uint32_t y = blockIdx.y;
    scan_local_body<MULT>(a.out[y], a.in[y], a.tot[y], n, a.reverse[y], a.exclusive[y]);
}

// phase 2: exclusive scan of the block totals by one workgroup (in place)
template <bool MULT>
__device__ __forceinline__ void scan_totals_body(Fr *tot, uint32_t nb) {
    __shared__ __attribute__((aligned(16))) Fr sh[1024];
    const uint32_t tid = threadIdx.x, per = (nb + 1023) / 1024;
    uint32_t lo = tid * per, hi = lo + per < nb ? lo + per : nb;
    if (lo > nb) lo = nb;
    Fr run = ident<MULT>();
    for (uint32_t i = lo; i < hi; i++) run = op<MULT>(run, load_fp(tot + i));
    sh[tid] = run;
    __syncthreads();
    Fr incl = run;
    for (int off = 1; off < 1024; off <<= 1) {
        Fr o = (int)tid >= off ? sh[tid - off] : ident<MULT>();
        __syncthreads();
        if ((int)tid >= off) { incl = op<MULT>(o, incl); sh[tid] = incl; }
        __syncthreads();
    }
    Fr acc = tid ? sh[tid - 1] : ident<MULT>();
    for (uint32_t i = lo; i < hi; i++) { Fr x = load_fp(tot + i); store_fp(tot + i, acc); acc = op<MULT>(acc, x); }
    if (lo < hi && hi == nb) store_fp(tot + nb, fin<MULT>(acc));         // the grand total, canonical, in the slot behind the nb prefixes
}

template <bool MULT> __global__ void __launch_bounds__(1024) k_scan_totals(Fr *tot, uint32_t nb) { scan_totals_body<MULT>(tot, nb); }
template <bool MULT> __global__ void __launch_bounds__(1024) k_scan_totals_pair(ScanPair a, uint32_t nb) { scan_totals_body<MULT>(a.tot[blockIdx.y], nb); }

// phase 3: fold the block prefix in
template <bool MULT>
__device__ __forceinline__ void scan_apply_body(Fr *out, const Fr *tot, uint32_t n, int reverse) {
    if (blockIdx.x == 0) return;
    Fr pre = load_fp(tot + blockIdx.x);
    const uint32_t base = blockIdx.x * BLOCK_ELEMS + threadIdx.x * EPT;
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        uint32_t i = base + k;
        if (i < n) { Fr *p = out + (reverse ? n - 1 - i : i); store_fp(p, fin<MULT>(op<MULT>(pre, load_fp(p)))); }
    }
}

template <bool MULT> __global__ void __launch_bounds__(PT) k_scan_apply(Fr *out, const Fr *tot, uint32_t n, int reverse) { scan_apply_body<MULT>(out, tot, n, reverse); }
template <bool MULT> __global__ void __launch_bounds__(PT) k_scan_apply_pair(ScanPair a, uint32_t n) {
    scan_apply_body<MULT>(a.out[blockIdx.y], a.tot[blockIdx.y], n, a.reverse[blockIdx.y]);
}

// two product scans of length n at once: (out0 <- scan of in0, reverse0, exclusive0) and the same for 1
// pre0 / pre1 (optional): leave the third phase to the consumer (mul3_blocks) — on return *pre points at the scan's nb block prefixes (scan order)
// followed by its grand total (canonical), valid until the context's next scan; nullptr when the scans fit one block (then they are complete).
int32_t scan_pair_mult(plk_ctx *ctx, Fr *out0, const Fr *in0, bool reverse0, bool exclusive0, Fr *out1, const Fr *in1, bool reverse1, bool exclusive1,
                       uint32_t n, hipStream_t s, const Fr **pre0, const Fr **pre1) {
    const uint32_t nb = (n + BLOCK_ELEMS - 1) / BLOCK_ELEMS;
    PLK_TRY(ctx->poly_tmp.reserve((size_t)2 * (nb + 1) * sizeof(Fr)));
    ScanPair a;
    a.out[0] = out0; a.in[0] = in0; a.reverse[0] = reverse0 ? 1 : 0; a.exclusive[0] = exclusive0 ? 1 : 0; a.tot[0] = ctx->poly_tmp.as<Fr>();
    a.out[1] = out1; a.in[1] = in1; a.reverse[1] = reverse1 ? 1 : 0; a.exclusive[1] = exclusive1 ? 1 : 0; a.tot[1] = ctx->poly_tmp.as<Fr>() + nb + 1;
    const bool defer = pre0 && pre1 && nb > 1;
    if (pre0) *pre0 = defer ? a.tot[0] : nullptr;
    if (pre1) *pre1 = defer ? a.tot[1] : nullptr;
    hipLaunchKernelGGL(k_scan_local_pair<true>, dim3(nb, 2), dim3(PT), 0, s, a, n);
    if (nb > 1) {
        hipLaunchKernelGGL(k_scan_totals_pair<true>, dim3(1, 2), dim3(1024), 0, s, a, nb);
        if (!defer) hipLaunchKernelGGL(k_scan_apply_pair<true>, dim3(nb, 2), dim3(PT), 0, s, a, n);
    }
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

int32_t scan(plk_ctx *ctx, Fr *out, const Fr *in, uint32_t n, bool mult, bool reverse, bool exclusive, hipStream_t s, DevBuf *totals) {
    uint32_t nb = (n + BLOCK_ELEMS - 1) / BLOCK_ELEMS;
    DevBuf &tb = totals ? *totals : ctx->poly_tmp;
    PLK_TRY(tb.reserve((size_t)(nb + 1) * sizeof(Fr)));           // nb block totals + the grand total (scan_totals_body)
    Fr *tot = tb.as<Fr>();
    if (mult) {
        hipLaunchKernelGGL(k_scan_local<true>, dim3(nb), dim3(PT), 0, s, out, in, tot, n, reverse ? 1 : 0, exclusive ? 1 : 0);
        if (nb > 1) {
            hipLaunchKernelGGL(k_scan_totals<true>, dim3(1), dim3(1024), 0, s, tot, nb);
            hipLaunchKernelGGL(k_scan_apply<true>, dim3(nb), dim3(PT), 0, s, out, (const Fr *)tot, n, reverse ? 1 : 0);
        }
    } else {
        hipLaunchKernelGGL(k_scan_local<false>, dim3(nb), dim3(PT), 0, s, out, in, tot, n, reverse ? 1 : 0, exclusive ? 1 : 0);
        if (nb > 1) {
            hipLaunchKernelGGL(k_scan_totals<false>, dim3(1), dim3(1024), 0, s, tot, nb);
            hipLaunchKernelGGL(k_scan_apply<false>, dim3(nb), dim3(PT), 0, s, out, (const Fr *)tot, n, reverse ? 1 : 0);
        }
    }
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

// ------------------------------------------------------------------- witness temporaries
struct DevWitnessTerm { uint32_t var, pad; uint32_t coeff[8]; };      // == circuit.h WitnessTerm (uint32 + HFr, 8-byte aligned)
struct DevWitnessOp { uint32_t first, count; uint32_t constant[8]; }; // == circuit.h WitnessOp
__device__ __forceinline__ Fr fr_of(const uint32_t *w) { Fr r; for (int i = 0; i < 8; i++) r.l[i] = w[i]; return r; }
static_assert(sizeof(DevWitnessTerm) == 40 && sizeof(DevWitnessOp) == 40, "layout of the uploaded records");
__global__ void __launch_bounds__(PT) k_eval_witness_ops(Fr *values, const DevWitnessOp *ops, const DevWitnessTerm *terms, uint32_t n_ops, uint32_t first_tmp) {
    uint32_t i = blockIdx.x * PT + threadIdx.x;
    if (i >= n_ops) return;
    const uint32_t first = ops[i].first, count = ops[i].count;
    Fr acc = fr_of(ops[i].constant);
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t var = terms[first + k].var;
        if (var) acc = add(acc, mul(fr_of(terms[first + k].coeff), load_fp(values + var)));      // id 0 is the dummy (zero)
    }
    store_fp(values + first_tmp + i, acc);
}
// The temporaries of a long linear combination form a CHAIN (partial sums handed from gate to gate through d):
// temporary i may read temporary i - 1 and nothing else that is not a circom wire.  One lane walks one such run from its head;
// runs are independent of each other, so the whole table is still one launch (a 2^20-gate Poseidon-shaped circuit: 0.7 M
// temporaries in runs of 1..20 — on the host this loop was 70 ms of every proof).
__global__ void __launch_bounds__(PT) k_eval_witness_runs(Fr *values, const DevWitnessOp *ops, const DevWitnessTerm *terms, const uint32_t *run_start,
                                                          uint32_t n_runs, uint32_t n_ops, uint32_t first_tmp) {
    uint32_t r = blockIdx.x * PT + threadIdx.x;
    if (r >= n_runs) return;
    const uint32_t lo = run_start[r], hi = r + 1 < n_runs ? run_start[r + 1] : n_ops;
    Fr prev; for (int k = 0; k < 8; k++) prev.l[k] = 0;
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t first = ops[i].first, count = ops[i].count;
        Fr acc = fr_of(ops[i].constant);
        for (uint32_t k = 0; k < count; k++) {
            const uint32_t var = terms[first + k].var;
            if (!var) continue;                                                       // id 0 is the dummy (zero)
            const Fr v = (i > lo && var == first_tmp + i - 1) ? prev : load_fp(values + var);
            acc = add(acc, mul(fr_of(terms[first + k].coeff), v));
        }
        store_fp(values + first_tmp + i, acc);
        prev = acc;
    }
}
int32_t eval_witness_runs(Fr *values, const void *ops_dev, const void *terms_dev, const void *run_start_dev, uint32_t n_runs, uint32_t n_ops, uint32_t first_tmp, hipStream_t s) {
    if (!n_ops) return PLK_OK;
    hipLaunchKernelGGL(k_eval_witness_runs, dim3((n_runs + PT - 1) / PT), dim3(PT), 0, s, values, (const DevWitnessOp *)ops_dev, (const DevWitnessTerm *)terms_dev,
                       (const uint32_t *)run_start_dev, n_runs, n_ops, first_tmp);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t eval_witness_ops(Fr *values, const void *ops_dev, const void *terms_dev, uint32_t n_ops, uint32_t first_tmp, hipStream_t s) {
    if (!n_ops) return PLK_OK;
    hipLaunchKernelGGL(k_eval_witness_ops, dim3((n_ops + PT - 1) / PT), dim3(PT), 0, s, values, (const DevWitnessOp *)ops_dev, (const DevWitnessTerm *)terms_dev, n_ops, first_tmp);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

// ----------------------------------------------------------------------------- quotient

// out_i = in_i * c as ONE product of the 29-bit layer (c given with the scale the caller wants, see QuotientArgs)
__global__ void __launch_bounds__(PT) k_scale_const(Fr *out, const Fr *in, Fr c, uint32_t n) {
    uint32_t i = blockIdx.x * PT + threadIdx.x;
    if (i < n) store_fp(out + i, pack<FrParams>(csub_p(mulw(ldw(in + i), cw(c)))));
}
// out_i = c * omega_m^j, j = the natural index of coset-major position i (k = i / n, r = i % n, j = 4 r + k)
// (tw_w: the power table of omega_{2^28} in the 2^261 domain)
__global__ void __launch_bounds__(PT) k_coset_points_w(Fr *out, PowTable tw_w, uint32_t shift, Fr c, uint32_t m, uint32_t log_n) {
    uint32_t i = blockIdx.x * PT + threadIdx.x;
    if (i >= m) return;
    const uint32_t j = ((i & ((1u << log_n) - 1)) << 2) | (i >> log_n);
    const uint32_t e = j << shift;
    const FrW9 w = mulw(ldw(tw_w.lo + (e & (POW_TAB - 1))), ldw(tw_w.hi + (e >> POW_SPLIT)));
    store_fp(out + i, pack<FrParams>(csub_p(mulw(w, cw(c)))));
}

// t(x) = [gate + PI + alpha*(perm) + alpha^2*L0*(z-1)] / Z_H(x) on the coset 7*<omega_4N>.
// 24 products per point on the 29-bit layer, the seven gate products in two fused sums (one reduction per three).
// Every vector — the inputs and the quotient itself — is in the COSET-MAJOR layout of lde4cm_batch_dev: position i = k * N + r
// holds the value at x_j = 7 * omega_4N^j, j = 4 r + k.  f(omega * x) is then the next row of the same coset, 1 / Z_H depends
// on k only; the coset iNTT that follows (icoset4cm_dev + k_icoset_combine) takes this order.
__global__ void __launch_bounds__(PT) k_quotient(QuotientArgs a) {
    const uint32_t log_n = a.log_m - 2, nmask = (1u << log_n) - 1;
    const uint32_t i = blockIdx.x * PT + threadIdx.x;
    if (i >= a.m) return;
    const uint32_t kc = i >> log_n, r = i & nmask;
    const uint32_t jnat = (r << 2) | kc;
    const uint32_t nxt = (kc << log_n) | ((r + 1) & nmask);      // f(omega*x): the next row of this coset
    FrW9 w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = ldw(a.w[j] + i);
    const FrW9 w01 = mulw(w[0], w[1]);                           // scale 2^251; q_m is stored with 2^266
    const FrW9 f1 = mulsum3w(ldw(a.q[0] + i), w[0], ldw(a.q[1] + i), w[1], ldw(a.q[2] + i), w[2]);
    const FrW9 f2 = mulsum3w(ldw(a.q[3] + i), w[3], ldw(a.q[4] + i), w01, ldw(a.q[6] + i), ldw(a.w[3] + nxt));
    FrW9 g = addn(addn(f1, f2), ldw(a.q[5] + i));
    if (a.pi) g = addn(g, ldw(a.pi + i));
    else for (uint32_t k = 0; k < a.num_pi; k++) g = addn(g, mulw(ldw(a.l0 + ((kc << log_n) | ((r - k) & nmask))), cw(a.pi_in[k])));
    FrW9 x;
    if (a.x) x = ldw(a.x + i);
    else {                                                        // no cached coset points (largest domains): 7 * omega_4N^i from the table
        const uint32_t e = jnat << (MAX_LOG_N - a.log_m);
        x = mulw(mulw(ldw(a.tw_w.lo + (e & (POW_TAB - 1))), ldw(a.tw_w.hi + (e >> POW_SPLIT))), cw(a.coset_w));
    }
    const FrW9 z = ldw(a.z + i), gamma = cw(a.gamma);
    // the per-proof constants as mul_tw3 operands (uniform: they sit in scalar registers); a mul_tw3 result is < 4p for a normalised operand —
    // every use below adds it to something and normalises (well inside the 2^261 capacity), the last product of the kernel stays a mulw
    Tw3<FrW> beta3, alpha_pp3, alpha2_3;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int l = 0; l < 9; l++) { beta3.w[q][l] = a.beta3[9 * q + l]; alpha_pp3.w[q][l] = a.alpha_pp3[9 * q + l]; alpha2_3.w[q][l] = a.alpha2_3[9 * q + l]; }
    FrW9 pa = z, pb = ldw(a.z + nxt);
    // beta * k_j * x for the coset representatives k = (1, 5, 7, 10) (prover.hip NON_RESIDUES): ONE product
    // beta * x and three small multiples formed limb-wise (5 = 4 + 1, 7 = 8 - 1, 10 = 2 * 5; limbs stay below 2^32) instead
    // of four products
    FrW9 bkx[4];
    bkx[0] = mul_tw3(x, beta3);                                  // (< 4p; the multiples below < 40p)
    {
        FrW9 t5, t7;
#pragma unroll
        for (int l = 0; l < 9; l++) { t5.l[l] = (bkx[0].l[l] << 2) + bkx[0].l[l]; t7.l[l] = (bkx[0].l[l] << 3) - bkx[0].l[l]; }
        bkx[1] = normw(t5); bkx[2] = normw(t7);
        FrW9 t10;
#pragma unroll
        for (int l = 0; l < 9; l++) t10.l[l] = bkx[1].l[l] << 1;
        bkx[3] = normw(t10);
    }
    static_for<4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const FrW9 wg = addn(w[j], gamma);
        pa = mulw(pa, addn(wg, bkx[j]));
        pb = mulw(pb, addn(wg, mul_tw3(ldw(a.sigma[j] + i), beta3)));
    });
    FrW9 t = addn(g, mul_tw3(sub2(pa, pb), alpha_pp3));
    t = addn(t, mul_tw3(mulw(ldw(a.l0 + i), sub2(z, cw(Fr::one()))), alpha2_3));
    Fr zh_inv = a.zh_inv_w[0];
#pragma unroll
    for (uint32_t c = 1; c < 4; c++) if (kc == c) zh_inv = a.zh_inv_w[c];
    store_fp(a.out + i, pack<FrParams>(csub_p(mulw(t, cw(zh_inv)))));
}

// Second half of the coset iNTT at 4N from the coset-major layout.  t(x) = sum_c x^(cN) T_c(x) with deg T_c < N; on the coset
// g_k * <omega_N> (g_k = 7 * omega_4N^k) x^N = 7^N * i^k (i = omega_4), so the polynomial u_k interpolated there (icoset4cm_dev) is
// u_k = sum_c (7^N i^k)^c T_c and, coefficient by coefficient,  T_c = 7^(-Nc) * (1/4) * sum_k i^(-kc) u_k : a 4-point inverse DFT
// (one product by i^-1) and four constant scalings, in place (thread j owns the four slots {cN + j}).  Output: natural
// coefficient order, canonical — what the three-pass 4N transform produced, at ~60 % of its cost together with the first half.
__global__ void __launch_bounds__(PT) k_icoset_combine(Fr *data, uint32_t n, Fr iinv_w, Fr s0_w, Fr s1_w, Fr s2_w, Fr s3_w) {
    const uint32_t j = blockIdx.x * PT + threadIdx.x;
    if (j >= n) return;
    const FrW9 u0 = ldw(data + j), u1 = ldw(data + n + j), u2 = ldw(data + 2 * (size_t)n + j), u3 = ldw(data + 3 * (size_t)n + j);
    const FrW9 a = addn(u0, u2), b = sub2(u0, u2), c = addn(u1, u3);
    const FrW9 d = mulw(sub2(u1, u3), cw(iinv_w));                                       // (u1 - u3) * i^-1
    stw(data + j, csub_p(mulw(addw(a, c), cw(s0_w))));
    stw(data + n + j, csub_p(mulw(addw(b, d), cw(s1_w))));
    stw(data + 2 * (size_t)n + j, csub_p(mulw(sub4(a, c), cw(s2_w))));
    stw(data + 3 * (size_t)n + j, csub_p(mulw(sub2(b, d), cw(s3_w))));
}
int32_t icoset_combine(Fr *data, uint32_t n, const Fr &iinv_w, const Fr s_w[4], hipStream_t s) {
    hipLaunchKernelGGL(k_icoset_combine, dim3((n + PT - 1) / PT), dim3(PT), 0, s, data, n, iinv_w, s_w[0], s_w[1], s_w[2], s_w[3]);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

// ------------------------------------------------------------------- linear combinations
// out_i = sum_k s_k * p_k[i]; a.s[k] = W(s_k) (unit[k]: the coefficient is one, the term is only added).  Two products share
// one Montgomery reduction (mul2addw: 243 multiply-adds instead of 324).
__global__ void __launch_bounds__(PT) k_lincomb(LinCombArgs a) {
    uint32_t i = blockIdx.x * PT + threadIdx.x;
    if (i >= a.n) return;
    FrW9 acc = w_zero<FrW>();
    FrW9 pv = w_zero<FrW>(), ps = w_zero<FrW>();
    bool pending = false;
    for (uint32_t k = 0; k < a.count; k++) {
        const FrW9 v = ldw(a.p[k] + i);
        if (a.unit[k]) { acc = addn(acc, v); continue; }
        const FrW9 sk = cw(a.s[k]);
        if (!pending) { pv = v; ps = sk; pending = true; }
        else { acc = addn(acc, mul2addw(pv, ps, v, sk)); pending = false; }
    }
    if (pending) acc = addn(acc, mulw(pv, ps));
    if (a.times_pow.lo) stw(a.out + i, csub_p(mulw(normw(acc), pow2l_w_(a.times_pow, i))));   // (k_mul_powers folded in)
    else stw(a.out + i, reduce_small(acc));                                             // < 16 p here
}

// out_i = in_i * base^(i + shift)     (base given by its W-domain power table)
__global__ void __launch_bounds__(PT) k_mul_powers(Fr *out, const Fr *in, PowTable t, uint32_t shift, uint32_t n) {
    uint32_t i = blockIdx.x * PT + threadIdx.x;
    if (i < n) stw(out + i, csub_p(mulw(ldw(in + i), pow2l_w_(t, i + shift))));
}

// q_k = S_{k+1} * zinv^(k+1), q_{n-1} = 0       (synthetic division by (x - z), see prover.hip)
__global__ void __launch_bounds__(PT) k_div_finish(Fr *q, const Fr *suffix, PowTable zinv, uint32_t n) {
    uint32_t k = blockIdx.x * PT + threadIdx.x;
    if (k >= n) return;
    if (k == n - 1) { store_fp(q + k, Fr::zero()); return; }
    stw(q + k, csub_p(mulw(ldw(suffix + k + 1), pow2l_w_(zinv, k + 1))));
}

// ------------------------------------------------------------------------- evaluation
// partial[b] = sum over the block's BLOCK_ELEMS coefficients c_i * x^i; Horner inside a thread with W(x) (the raw sum
// acc * x + c feeds the next product's left side un-normalised), then one product by W(x^start); canonical before the
// 8 x 32-bit tree sum
__global__ void __launch_bounds__(PT) k_eval_partial(EvalArgs a) {
    __shared__ __attribute__((aligned(16))) Fr sh[PT];
    const uint32_t e = blockIdx.y, tid = threadIdx.x;
    const uint32_t n = a.len[e];
    const uint32_t start = blockIdx.x * BLOCK_ELEMS + tid * EPT;
    Fr acc = Fr::zero();
    if (start < n) {
        const Fr *c = a.poly[e];
        const FrW9 x = ldw(a.pt[e].lo + 1);
        uint32_t hi = start + EPT < n ? start + EPT : n;
        FrW9 h = ldw(c + hi - 1);
        for (int i = (int)hi - 2; i >= (int)start; i--) h = addw(mulw(h, x), ldw(c + i));
        acc = pack<FrParams>(csub_p(mulw(normw(h), pow2l_w_(a.pt[e], start))));
    }
    sh[tid] = acc;
    __syncthreads();
    for (int off = PT / 2; off > 0; off >>= 1) {
        if ((int)tid < off) { acc = add(acc, sh[tid + off]); sh[tid] = acc; }
        __syncthreads();
    }
    if (tid == 0) store_fp(a.partials + (size_t)e * a.max_blocks + blockIdx.x, acc);
}

__global__ void __launch_bounds__(PT) k_eval_finish(EvalArgs a, Fr *results) {
    __shared__ __attribute__((aligned(16))) Fr sh[PT];
    const uint32_t e = blockIdx.x, tid = threadIdx.x;
    const uint32_t nb = (a.len[e] + BLOCK_ELEMS - 1) / BLOCK_ELEMS;
    Fr acc = Fr::zero();
    for (uint32_t b = tid; b < nb; b += PT) acc = add(acc, load_fp(a.partials + (size_t)e * a.max_blocks + b));
    sh[tid] = acc;
    __syncthreads();
    for (int off = PT / 2; off > 0; off >>= 1) {
        if ((int)tid < off) { acc = add(acc, sh[tid + off]); sh[tid] = acc; }
        __syncthreads();
    }
    if (tid == 0) store_fp(results + e, acc);
}

// ------------------------------------------------------------------- gate satisfiability
// is_satisfied_using_one_shot_check (src/plonk.rs:128,137) on the device: every row's gate equation
// q_a a + q_b b + q_c c + q_d d + q_m ab + q_const + q_dnext d_next + PI == 0; flag <- 1 otherwise
__global__ void __launch_bounds__(PT) k_check_gates(CheckArgs a) {
    uint32_t r = blockIdx.x * PT + threadIdx.x;
    if (r >= a.n) return;
    Fr w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = load_fp(a.values + a.vars[j][r]);
    Fr acc = load_fp(a.q[5] + r);
#pragma unroll
    for (int j = 0; j < 4; j++) acc = add(acc, mul(load_fp(a.q[j] + r), w[j]));
    acc = add(acc, mul(load_fp(a.q[4] + r), mul(w[0], w[1])));
    Fr qn = load_fp(a.q[6] + r);
    if (!qn.is_zero()) {
        Fr dn = (r + 1 < a.n) ? load_fp(a.values + a.vars[3][r + 1]) : Fr::zero();
        acc = add(acc, mul(qn, dn));
    }
    if (r < a.num_inputs) acc = add(acc, w[0]);
    if (!acc.is_zero()) atomicOr(a.flag, 1u);
}

// ------------------------------------------------------------------------- launchers
static inline dim3 grid1(uint32_t n) { return dim3((n + PT - 1) / PT); }

int32_t gather(Fr *out, const Fr *values, const uint32_t *vars, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_gather, grid1(n), dim3(PT), 0, s, out, values, vars, n);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t gather4_dual(Fr *const out[4], Fr *const copy[4], const Fr *values, const uint32_t *const vars[4], uint32_t n, hipStream_t s) {
    Gather4 a;
    for (int j = 0; j < 4; j++) { a.out[j] = out[j]; a.copy[j] = copy[j]; a.vars[j] = vars[j]; }
    hipLaunchKernelGGL(k_gather4, dim3((n + PT - 1) / PT, 4), dim3(PT), 0, s, a, values, n);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t sigma_from_index(Fr *out, const uint32_t *packed, uint32_t n, uint32_t log_n, const PowTable &tw, const Fr k[4], hipStream_t s) {
    hipLaunchKernelGGL(k_sigma_from_index, grid1(n), dim3(PT), 0, s, out, packed, n, log_n, tw, k[0], k[1], k[2], k[3]);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t check_gates(const CheckArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_check_gates, grid1(a.n), dim3(PT), 0, s, a);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t perm_terms(const PermArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_perm_terms, grid1(a.n), dim3(PT), 0, s, a);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t mul3_blocks(Fr *out, const Fr *a, const Fr *b, const Fr *pre_a, const Fr *pre_b, const Fr &sc, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_mul3_blocks, grid1(n), dim3(PT), 0, s, out, a, b, pre_a, pre_b, sc, n);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t mul3(Fr *out, const Fr *a, const Fr *b, const Fr &sc, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_mul3, grid1(n), dim3(PT), 0, s, out, a, b, sc, n);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t scale_const(Fr *out, const Fr *in, const Fr &c_s, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_scale_const, grid1(n), dim3(PT), 0, s, out, in, c_s, n);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t coset_points_w(Fr *out, const PowTable &tw_w, uint32_t log_m, const Fr &c_s, uint32_t m, hipStream_t s) {
    hipLaunchKernelGGL(k_coset_points_w, grid1(m), dim3(PT), 0, s, out, tw_w, MAX_LOG_N - log_m, c_s, m, log_m - 2);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t quotient(const QuotientArgs &args, hipStream_t s) {
    QuotientArgs a = args;
    // the three shifted copies of each per-proof constant (make_tw3 runs on the host: six products of the 29-bit layer)
    auto fill3 = [](uint32_t (&out)[27], const Fr &c) {
        const Tw3<FrW> t = make_tw3(unpack<FrW>(c));
        for (int q = 0; q < 3; q++) for (int l = 0; l < 9; l++) out[9 * q + l] = t.w[q][l];
    };
    fill3(a.beta3, a.beta); fill3(a.alpha_pp3, a.alpha_pp); fill3(a.alpha2_3, a.alpha2_w);
    hipLaunchKernelGGL(k_quotient, grid1(a.m), dim3(PT), 0, s, a);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t lincomb(const LinCombArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_lincomb, grid1(a.n), dim3(PT), 0, s, a);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t mul_powers(Fr *out, const Fr *in, const PowTable &t, uint32_t shift, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_mul_powers, grid1(n), dim3(PT), 0, s, out, in, t, shift, n);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t div_finish(Fr *q, const Fr *suffix, const PowTable &zinv, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_div_finish, grid1(n), dim3(PT), 0, s, q, suffix, zinv, n);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}
int32_t eval_batch(plk_ctx *ctx, EvalArgs a, Fr *results_dev, hipStream_t s) {
    uint32_t maxlen = 0;
    for (uint32_t e = 0; e < a.count; e++) if (a.len[e] > maxlen) maxlen = a.len[e];
    a.max_blocks = (maxlen + BLOCK_ELEMS - 1) / BLOCK_ELEMS;
    PLK_TRY(ctx->poly_tmp2.reserve((size_t)a.max_blocks * a.count * sizeof(Fr)));
    a.partials = ctx->poly_tmp2.as<Fr>();
    hipLaunchKernelGGL(k_eval_partial, dim3(a.max_blocks, a.count), dim3(PT), 0, s, a);
    hipLaunchKernelGGL(k_eval_finish, dim3(a.count), dim3(PT), 0, s, a, results_dev);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

}  // namespace plk
