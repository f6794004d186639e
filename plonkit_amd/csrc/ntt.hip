// Radix-2 NTT / iNTT / coset NTT over BN254 Fr for gfx950 — natural order in, natural order out.
//
// Replaces bellman_ce's Polynomial::{fft, ifft, coset_fft, icoset_fft} / fft::best_fft, which the
// reference drives from setup() (src/plonk.rs:104, 11 iNTT(N)) and prove_by_steps
// (src/plonk.rs:152-159: 6 iNTT(N), ~18 coset NTT(4N), 1 coset iNTT(4N)).  omega_n, ordering and
// the coset generator 7 are those of SURVEY.md A.2/A.4.
//
// Design (MI355X-first, not a translation of bellman's thread-split FFT):
//   * n = 2^log_n is factored into p <= 3 digits of <= 10 bits: n = R1*R2*..*Rp (mixed-radix Cooley-Tukey,
//     "four-step" generalised): two passes up to 2^20, three beyond.  Pass i transforms digit i for every
//     combination of the other digits; between passes the element (k_i, m) is multiplied by omega_L^(k_i*m).
//   * A workgroup (512 threads) owns a tile of 2048 elements = R rows x C adjacent columns (C*32 B contiguous
//     per row), stages it in LDS as 9 x 29-bit limbs in two 16-byte planes and one 4-byte plane, and runs the
//     log2(R) butterfly stages there two at a time (radix 4 in registers): HBM sees exactly one read and one
//     write of the vector per pass.
//   * Every pass is DIT.  Passes 1..p-1 fetch their rows in bit-reversed order (a row is its own memory segment,
//     so the order is free); the last pass reads contiguous rows, bit-reverses them with its LDS scatter and
//     writes whole C-element segments to the digit-reversed output index.
//   * All butterfly arithmetic is the carry-free 9 x 29-bit lazy layer (field29_dev.h): data words are re-sliced,
//     never converted; only the constants (twiddles, coset powers, 1/n) live in its 2^261 domain.
//   * coset shift (g^i on load), 1/n and g^-i (on store) are fused into the first / last pass; powers come
//     from two-level tables (base^(lo + 2^14*hi)), one multiply per use — none when the low part is zero
//     (sub-transforms of <= 2^14 points).  The twiddles between the passes of a larger transform are read from
//     a table built per (direction, digit plan) at first use (ntt_direct_table below; 1/n folded in).  A
//     zero-padded input (LDE) is read in place: indices beyond the coefficient count are neither loaded nor
//     scaled, and the first pair of stages of its first pass is a copy.  Outputs are made canonical by a
//     quotient-estimate reduction, not by a product.
//   * Up to 16 transforms of one shape share every launch (grid.y, per-transform input tables): the four wire iNTTs of a
//     proof, and the coset-major extension lde4cm_batch_dev — 4n evaluations as four n-point coset transforms (cosets
//     7*omega_4n^k * <omega_n>) per polynomial, the layout the quotient kernel reads.  Two scratch lanes, so that transforms on the
//     prover's background stream do not share the ping-pong buffer with the main stream's.
//   * No MFMA: this is 256-bit modular integer arithmetic, bound by v_mad_u64_u32 issue.
#include "ctx.h"
#include "ntt.h"
#include "poly.h"
#include "field29_dev.h"
#include "ntt_plan.h"
#include <cstdlib>
#include <algorithm>
#include <vector>

namespace plk {

constexpr int NTT_THREADS = 512;               // 8 waves per workgroup, 2 workgroups per CU (LDS): 4 waves per SIMD
constexpr int LOG_TILE = 11;                   // 2048 elements per workgroup
constexpr int LOG_SINGLE = 11;                 // largest transform done by one workgroup in one pass (ntt_pass_rows)

constexpr uint32_t NTT_MAX_BATCH = 16;         // transforms of equal shape sharing one launch per pass (blockIdx.y)
struct NttPassArgs {
    const Fr *in_b[NTT_MAX_BATCH];             // per transform of the batch: where this pass reads ...
    Fr *out_b[NTT_MAX_BATCH];                  // ... and writes
    uint32_t log_n, log_r, log_c;
    uint32_t log_inner;                        // type-A: row stride = 2^log_inner
    uint32_t log_r1, log_m1, log_m2;           // final pass: digit widths of k1 and the middle digits
    PowTable tw;                               // omega_{2^28}^(+-e)
    PowTable pre_b[NTT_MAX_BATCH];             // optional, per transform: multiply input i by pre^i (first pass)
    PowTable post_b[NTT_MAX_BATCH];            // optional, per transform: multiply output k by post^k (last pass)
    Fr scale;                                  // optional 1/n on the last pass
    uint32_t has_scale;
    uint32_t nonzero;                          // first pass: input elements at index >= nonzero are zero and are
                                               // neither read nor scaled (zero-padded LDE); 0 = the whole vector
    const Fr *tw_direct;                       // optional (ntt_pass_cols): the inter-pass twiddles of this pass as a table,
                                               // entry (k << log_inner | column), W domain, 1/n folded in on inverse transforms
    uint32_t quarter;                          // first pass of an LDE by 4 with an even number of stages: only every fourth
                                               // (bit-reversed) row is non-zero, the first pair of stages is a broadcast
    const Fr *pre_direct_b[NTT_MAX_BATCH];     // optional, per transform: pre^i as a table of its own (W domain), index = element index:
    const Fr *post_direct_b[NTT_MAX_BATCH];    // one load + one product per element instead of two loads + two products (the composition
};                                             // lo[e & 16383] * hi[e >> 14] is itself a product) — the coset shifts of the prover's extensions

// ---- all butterfly arithmetic runs on the carry-free 9 x 29-bit layer (field29_dev.h): data words are
// ---- re-sliced, never converted; the constants (twiddles, coset powers, 1/n) live in its 2^261 domain.
__device__ __forceinline__ FrW9 ldw(const Fr *p) { return unpack<FrW>(load_fp(p)); }

__device__ __forceinline__ FrW9 pow2l_w(const PowTable &t, uint32_t e) {
    return mulw(ldw(t.lo + (e & (POW_TAB - 1))), ldw(t.hi + (e >> POW_SPLIT)));
}

__device__ __forceinline__ uint32_t brev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// LDS tile: 9 limbs per element as two 16-byte planes and one 4-byte plane
struct LdsTile {
    u32x4 *p0, *p1;
    uint32_t *p2;
    __device__ __forceinline__ FrW9 get(uint32_t i) const {
        u32x4 a = p0[i], b = p1[i];
        FrW9 r;
        r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
        r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = p2[i];
        return r;
    }
    __device__ __forceinline__ void put(uint32_t i, const FrW9 &v) const {
        p0[i] = u32x4{v.l[0], v.l[1], v.l[2], v.l[3]};
        p1[i] = u32x4{v.l[4], v.l[5], v.l[6], v.l[7]};
        p2[i] = v.l[8];
    }
};

// omega_R^i (i < R/2) straight out of the hi table: omega_{2^28}^(i << (28 - log_r)), low part 0
// (read from the pre-sliced copy of the table: 3 x 16 bytes, no 8x32 -> 9x29 re-slicing — the butterfly kernels
//  are bound by instruction issue, and the re-slicing was ~7 % of a radix-4 group)
__device__ __forceinline__ FrW9 small_tw(const PowTable &t, uint32_t i, uint32_t log_r) {
    const u32x4 *p = reinterpret_cast<const u32x4 *>(t.hi_sliced) + 3 * (size_t)(i << (POW_SPLIT - log_r));
    const u32x4 a = p[0], b = p[1], c = p[2];
    FrW9 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = c.x;
    return r;
}

// log2(R) DIT stages over an LDS tile of R rows x C columns (element (i, c) at i*pitch + c), input rows in
// bit-reversed order.  Values stay lazily reduced: each stage adds at most 2p.
//
// Two stages at a time (radix 4 in registers): a group is the four rows i0 + {0, h, 2h, 3h}; stage s pairs
// (0,1) and (2,3) with omega_{2h}^jl, stage s+1 pairs (0,2) with omega_{4h}^jl and (1,3) with omega_{4h}^(jl+h).
// That halves the LDS round trips and barriers.  With 74 KB of LDS per tile two workgroups share a CU, so the
// workgroup is 512 threads (one group per thread and step, 118 VGPRs): four waves per SIMD hide the LDS and L2
// latencies.  Measured at 2^20: radix 2 / 256 threads 0.197 ms; radix 4 / 256 threads with two groups per thread
// loaded up front (196 VGPRs) 0.150 ms; radix 4 / 512 threads 0.142 ms.
struct Radix4Group {
    FrW9 x0, x1, x2, x3, t1, t2, t3;
    uint32_t a0, a1, a2, a3;
};

__device__ __forceinline__ void r4_load(Radix4Group &q, const LdsTile &L, const PowTable &tw, uint32_t g, uint32_t s,
                                        uint32_t log_r, uint32_t log_c, uint32_t pitch) {
    const uint32_t C = 1u << log_c, h = 1u << s;
    const uint32_t c = g & (C - 1), j = g >> log_c, jl = j & (h - 1);
    const uint32_t i0 = ((j >> s) << (s + 2)) | jl;
    q.a0 = i0 * pitch + c; q.a1 = q.a0 + h * pitch; q.a2 = q.a1 + h * pitch; q.a3 = q.a2 + h * pitch;
    q.x0 = L.get(q.a0); q.x1 = L.get(q.a1); q.x2 = L.get(q.a2); q.x3 = L.get(q.a3);
    if (s) q.t1 = small_tw(tw, jl << (log_r - s - 1), log_r);
    q.t2 = small_tw(tw, jl << (log_r - s - 2), log_r);
    q.t3 = small_tw(tw, (jl + h) << (log_r - s - 2), log_r);
}

// Lazy normalisation: mulw takes a left operand with limbs up to 3.28e9 (field29_dev.h, MULW_A_LIMB_MAX), so the sums and
// differences that only feed a product, or another sum, are left as raw limb-wise results; one carry propagation per
// OUTPUT (4 per group instead of 10).  Limb bounds (normalised = < 2^29 = 0.54e9; PAD2/PAD4 limbs < 2.68e9):
//     x2 + PAD2 - y3 < 3.22e9 (product operand)      x0 + PAD4 - y1 - b3 < 3.22e9      everything else smaller.
// Values grow by at most 4p per pair of stages, as before.
__device__ __forceinline__ void r4_finish(const Radix4Group &q, const LdsTile &L, uint32_t s) {
    // the two products of a stage are independent: issued in lockstep (mulw2, field29_dev.h)
    FrW9 y1 = q.x1, y3 = q.x3;                                                                      // stage 0: twiddle 1
    if (s) mulw2(q.x1, q.t1, q.x3, q.t1, y1, y3);
    FrW9 u, v;
#pragma unroll
    for (int i = 0; i < 9; i++) { u.l[i] = q.x2.l[i] + y3.l[i]; v.l[i] = q.x2.l[i] + FrW::PAD2[i] - y3.l[i]; }
    FrW9 b2 = u, b3;                                                                                // stage 1 of the first pair: omega_4^0 = 1
    if (s) mulw2(u, q.t2, v, q.t3, b2, b3); else b3 = mulw(v, q.t3);
    FrW9 o0, o1, o2, o3;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t b0 = q.x0.l[i] + y1.l[i];                                                    // x0 + y1
        o0.l[i] = b0 + b2.l[i];
        o2.l[i] = b0 + FrW::PAD4[i] - b2.l[i];                                                      // (b2 may be the raw x2 + y3 < 2.2p: 4p covers it)
        o1.l[i] = q.x0.l[i] + FrW::PAD2[i] - y1.l[i] + b3.l[i];
        o3.l[i] = q.x0.l[i] + FrW::PAD4[i] - y1.l[i] - b3.l[i];
    }
    L.put(q.a0, normw(o0)); L.put(q.a2, normw(o2));
    L.put(q.a1, normw(o1)); L.put(q.a3, normw(o3));
}

__device__ __forceinline__ void dit_stages(const LdsTile &L, const PowTable &tw, uint32_t log_r, uint32_t log_c, uint32_t pitch, uint32_t tid,
                                           bool quarter = false) {
    const uint32_t C = 1u << log_c, half_tile = 1u << (log_r + log_c - 1), quarter_tile = half_tile >> 1;
    uint32_t s = 0;
    if (quarter) {                                            // rows 4j+1..4j+3 hold zeros: stages 0 and 1 copy row 4j into them
        for (uint32_t g = tid; g < quarter_tile; g += NTT_THREADS) {
            const uint32_t c = g & (C - 1), j = g >> log_c;
            const uint32_t a0 = (j << 2) * pitch + c;
            const FrW9 x = L.get(a0);
            L.put(a0 + pitch, x); L.put(a0 + 2 * pitch, x); L.put(a0 + 3 * pitch, x);
        }
        __syncthreads();
        s = 2;
    }
    if (log_r & 1) {                                          // odd number of stages: one twiddle-free radix-2 stage first
        for (uint32_t b = tid; b < half_tile; b += NTT_THREADS) {
            uint32_t c = b & (C - 1), j = b >> log_c;
            uint32_t i0 = j << 1, i1 = i0 + 1;
            FrW9 u = L.get(i0 * pitch + c), v = normw(L.get(i1 * pitch + c));
            L.put(i0 * pitch + c, addn(u, v));
            L.put(i1 * pitch + c, sub2(u, v));               // inputs < 1.1p
        }
        __syncthreads();
        s = 1;
    }
    for (; s < log_r; s += 2) {
        for (uint32_t g = tid; g < quarter_tile; g += NTT_THREADS) {         // one iteration for a full 2048-element tile
            Radix4Group q;
            r4_load(q, L, tw, g, s, log_r, log_c, pitch);
            r4_finish(q, L, s);
        }
        __syncthreads();
    }
}

// Passes 1..p-1: strided columns, in place, post-multiplied by the inter-digit twiddle.
__global__ void __launch_bounds__(NTT_THREADS) ntt_pass_cols(NttPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t log_r = a.log_r, log_c = a.log_c, C = 1u << log_c;
    const uint32_t T = 1u << (log_r + log_c);
    LdsTile L{reinterpret_cast<u32x4 *>(smem), reinterpret_cast<u32x4 *>(smem) + T, reinterpret_cast<uint32_t *>(reinterpret_cast<u32x4 *>(smem) + 2 * T)};
    const uint32_t tid = threadIdx.x, t = blockIdx.x;
    const Fr *const in = a.in_b[blockIdx.y];
    Fr *const out = a.out_b[blockIdx.y];
    const PowTable pre = a.pre_b[blockIdx.y];
    const Fr *const pre_direct = a.pre_direct_b[blockIdx.y];
    const uint32_t tiles_log = a.log_inner - log_c;
    const uint32_t o = t >> tiles_log, c0 = (t & ((1u << tiles_log) - 1)) << log_c;
    const size_t base = ((size_t)o << (log_r + a.log_inner)) + c0;

    for (uint32_t idx = tid; idx < T; idx += NTT_THREADS) {
        uint32_t c = idx & (C - 1), p = idx >> log_c;
        size_t g = base + ((size_t)brev(p, log_r) << a.log_inner) + c;     // rows fetched in bit-reversed order
        FrW9 v = w_zero<FrW>();
        if (!a.nonzero || g < a.nonzero) {
            v = ldw(in + g);
            if (pre_direct) v = mulw(v, ldw(pre_direct + g));
            else if (pre.lo) v = mulw(v, pow2l_w(pre, (uint32_t)g));
        }
        L.put(idx, v);
    }
    __syncthreads();
    dit_stages(L, a.tw, log_r, log_c, C, tid, a.quarter != 0);
    const uint32_t eshift = MAX_LOG_N - (log_r + a.log_inner);
    for (uint32_t idx = tid; idx < T; idx += NTT_THREADS) {
        uint32_t c = idx & (C - 1), k = idx >> log_c;
        uint32_t e = (k * (c0 + c)) << eshift;
        // sub-transforms of <= 2^14 points only touch the hi table (the low 14 exponent bits are zero): no composition;
        // larger ones read the twiddle from the pass's own table when there is one (one load instead of a product)
        const FrW9 t = a.tw_direct ? ldw(a.tw_direct + ((size_t)k << a.log_inner) + c0 + c)
                     : eshift >= POW_SPLIT ? ldw(a.tw.hi + (e >> POW_SPLIT)) : pow2l_w(a.tw, e);
        FrW9 v = mulw(L.get(idx), t);                                         // < 1.1p: fits 256 bits, stays lazy
        store_fp(out + base + ((size_t)k << a.log_inner) + c, pack<FrParams>(v));
    }
}

// Last pass: C contiguous rows of R elements, scattered to the digit-reversed output index, canonical.
__global__ void __launch_bounds__(NTT_THREADS) ntt_pass_rows(NttPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t log_r = a.log_r, log_c = a.log_c, C = 1u << log_c, R = 1u << log_r;
    const uint32_t T = 1u << (log_r + log_c);
    LdsTile L{reinterpret_cast<u32x4 *>(smem), reinterpret_cast<u32x4 *>(smem) + T, reinterpret_cast<uint32_t *>(reinterpret_cast<u32x4 *>(smem) + 2 * T)};
    const uint32_t tid = threadIdx.x, t = blockIdx.x;
    const Fr *const in = a.in_b[blockIdx.y];
    Fr *const out = a.out_b[blockIdx.y];
    const PowTable pre = a.pre_b[blockIdx.y], post = a.post_b[blockIdx.y];
    const Fr *const pre_direct = a.pre_direct_b[blockIdx.y], *const post_direct = a.post_direct_b[blockIdx.y];
    const uint32_t kb_log = a.log_r1 - log_c, log_m = a.log_m1 + a.log_m2;
    const uint32_t k1_0 = (t & ((1u << kb_log) - 1)) << log_c, mu = t >> kb_log;

    for (uint32_t idx = tid; idx < T; idx += NTT_THREADS) {
        uint32_t n = idx & (R - 1), c = idx >> log_r;
        size_t rho = ((size_t)(k1_0 + c) << log_m) + mu;
        size_t g = (rho << log_r) + n;
        FrW9 v = w_zero<FrW>();
        if (!a.nonzero || g < a.nonzero) {
            v = ldw(in + g);
            if (pre_direct) v = mulw(v, ldw(pre_direct + g));
            else if (pre.lo) v = mulw(v, pow2l_w(pre, (uint32_t)g));
        }
        L.put(brev(n, log_r) * C + c, v);                                    // bit reversal as an LDS scatter
    }
    __syncthreads();
    dit_stages(L, a.tw, log_r, log_c, C, tid);
    const uint32_t k2 = mu >> a.log_m2, k3 = mu & ((1u << a.log_m2) - 1);
    const size_t drev = (size_t)k2 + ((size_t)k3 << a.log_m1);
    const FrW9 last = unpack<FrW>(a.scale);                                  // 1/n on inverse transforms
    for (uint32_t idx = tid; idx < T; idx += NTT_THREADS) {
        uint32_t c = idx & (C - 1), k = idx >> log_c;
        size_t o = (size_t)(k1_0 + c) + (drev << a.log_r1) + ((size_t)k << (a.log_n - log_r));
        FrW9 v = L.get(k * C + c);
        if (post_direct) v = mulw(v, ldw(post_direct + o));
        else if (post.lo) v = mulw(v, pow2l_w(post, (uint32_t)o));
        v = a.has_scale ? csub_p(mulw(v, last)) : reduce_small(v);          // canonical output (values here are < 24p)
        store_fp(out + o, pack<FrParams>(v));
    }
}

// ------------------------------------------------------------------------------- wave-owned tile passes (round 5)
// The same passes as ntt_pass_cols / ntt_pass_rows for full 2048-element tiles of 7..10 row bits, restructured so that the
// waves of a workgroup run apart (ntt_plan.h has the index plan and its host-side proof):
//   * round 0 takes its four rows straight from HBM, the last round multiplies by the inter-pass twiddle (or canonicalises)
//     and stores straight to HBM: NR - 1 LDS exchanges per pass instead of NR + 1;
//   * an exchange inside a phase is private to ONE wave (the LDS executes a wave's instructions in order, so a compiler
//     fence is all it needs); the only workgroup barriers of a pass are the two around the phase A -> phase B hand-over,
//     against one barrier per round (six per pass at 2^20) before.  Measured with rocprofv3 counters on the old kernels
//     (profiles/r05_ntt_pmc_before.txt): a wave was parked at s_waitcnt / s_barrier 38 % of its life, the VALU about 70 %
//     busy — the barriers made all eight waves of a tile (and usually both tiles of a CU) do their LDS traffic at the same time;
//   * LDS is laid out the way the reader reads it: 64 consecutive 16-byte words per instruction, no bank conflicts on either
//     side (pads per round from the plan), one address register with immediate offsets.
// The old kernels (ntt_pass_cols / ntt_pass_rows) keep the shapes this one does not take: transforms of <= 2^11 points, 6-bit passes
// (2^12, 2^13), and passes whose inter-pass twiddle table does not fit the cap.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// omega_R^i (i < R/2) as a mul_tw3 constant: 7 x 16 bytes out of the hi table's tw3 copy
__device__ __forceinline__ Tw3<FrW> small_tw3(const PowTable &t, uint32_t i, uint32_t log_r) {
    const u32x4 *p = reinterpret_cast<const u32x4 *>(t.hi_tw3) + 7 * (size_t)(i << (POW_SPLIT - log_r));
    uint32_t w[28];
#pragma unroll
    for (int k = 0; k < 7; k++) { const u32x4 v = p[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
    Tw3<FrW> r;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int k = 0; k < 9; k++) r.w[q][k] = w[9 * q + k];
    return r;
}

// stages S .. S+NST-1 on a lane's four rows row0 + {0, h, 2h, 3h} (h = 2^S); outputs normalised.
// The stage twiddles are mul_tw3 constants (108 multiply-adds per product instead of 162).  A mul_tw3 result is < 4p whatever the
// value of its (normalised) operand, so with inputs < V the outputs of a two-stage round are < V + 8p:
//     y1, y3 < 4p;   u = x2 + y3,  v = x2 + 4p - y3  (normalised before the second product);   b2, b3 < 4p;
//     o0 = x0 + y1 + b2,  o2 = x0 + y1 + 4p - b2,  o1 = x0 + 4p - y1 + b3,  o3 = x0 + 8p - y1 - b3        (all < V + 8p)
// and the first round (stage 0: y1 = x1, y3 = x3 < 2p, b2 = x2 + x3 un-multiplied) leaves < 7.3p from inputs < 1.3p.  Five rounds end
// below 40p: inside the 2^261 capacity (170p) and reduce_small's 64p.  Limbs: every sum below stays under 2^32 (x < 2^29, PADk limbs
// < 2.7e9, mul_tw3 outputs < 2^29), and the un-normalised operand of a product never occurs (u and v are normalised first).
// Where the twiddle loads sit is pinned (sched_barrier): left alone, the scheduler sinks every load down to just before its first use
// to save registers, and the wave then sits out one L2 latency (~750 cycles) per twiddle, three times per round.  t1 is requested
// BEFORE the LDS exchange that precedes the round (wave_round_t1), t2 before the first product, t3 right after it (when t1 is dead:
// four data elements + three 27-word twiddles would not fit 128 registers).
template <int LR, int S, int NST>
__device__ __forceinline__ Tw3<FrW> wave_round_t1(const PowTable &tw, uint32_t row0) {
    Tw3<FrW> t{};
    if constexpr (NST == 2 && S > 0) t = small_tw3(tw, (row0 & ((1u << S) - 1)) << (LR - S - 1), LR);
    return t;
}
template <int LR, int S, int NST>
__device__ __forceinline__ void wave_round(FrW9 &x0, FrW9 &x1, FrW9 &x2, FrW9 &x3, const PowTable &tw, uint32_t row0, const Tw3<FrW> &t1) {
    constexpr uint32_t h = 1u << S;
    if constexpr (NST == 1) {                                   // one twiddle-free stage (S = 0): pairs (0,1) and (2,3)
        static_assert(S == 0, "a single stage is only ever the first");
        const FrW9 v1 = normw(x1), v3 = normw(x3);
        const FrW9 u0 = x0, u2 = x2;
        x0 = addn(u0, v1); x1 = sub2(u0, v1);                  // inputs < 1.3p -> < 3.3p
        x2 = addn(u2, v3); x3 = sub2(u2, v3);
    } else {
        const uint32_t jl = row0 & (h - 1);
        FrW9 y1 = x1, y3 = x3;
        Tw3<FrW> t2{};
        if constexpr (S > 0) {
            t2 = small_tw3(tw, jl << (LR - S - 2), LR);
            __builtin_amdgcn_sched_barrier(0);
            mul_tw3_2(x1, x3, t1, y1, y3);
        }
        const Tw3<FrW> t3 = small_tw3(tw, (jl + h) << (LR - S - 2), LR);
        __builtin_amdgcn_sched_barrier(0);
        FrW9 u, v;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            u.l[i] = x2.l[i] + y3.l[i];
            v.l[i] = x2.l[i] + (S > 0 ? FrW::PAD4[i] : FrW::PAD2[i]) - y3.l[i];
        }
        FrW9 b2 = u;                                            // stage 1 of the first pair: omega_4^0 = 1
        if constexpr (S > 0) b2 = mul_tw3(normw(u), t2);
        const FrW9 b3 = mul_tw3(normw(v), t3);
        FrW9 o0, o1, o2, o3;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const uint32_t b0 = x0.l[i] + y1.l[i];
            o0.l[i] = b0 + b2.l[i];
            o2.l[i] = b0 + FrW::PAD4[i] - b2.l[i];             // (S = 0: b2 is the raw x2 + x3 < 2.6p)
            o1.l[i] = x0.l[i] + (S > 0 ? FrW::PAD4[i] : FrW::PAD2[i]) - y1.l[i] + b3.l[i];
            o3.l[i] = x0.l[i] + (S > 0 ? FrW::PAD8[i] : FrW::PAD6[i]) - y1.l[i] - b3.l[i];
        }
        x0 = normw(o0); x1 = normw(o1); x2 = normw(o2); x3 = normw(o3);
    }
}

// LT = 11: a 2048-element tile owned by eight waves, two workgroups per CU; LT = 12 (round 6): a 4096-element tile owned by sixteen waves, one workgroup
// per CU — 11-bit digits, so that a 2^21 / 2^22-point transform is two passes instead of three (ntt_plan.h).  Four waves per SIMD either way.
template <int LR, bool ROWS, int LT = NTT_LOG_TILE>
__global__ void __launch_bounds__(LT == NTT_LOG_TILE ? NTT_THREADS : 2 * NTT_THREADS, 4) ntt_pass_w(NttPassArgs a) {
    using P = TilePlan<LR, LT>;
    constexpr uint32_t LC = P::LC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LdsTile L{reinterpret_cast<u32x4 *>(smem), reinterpret_cast<u32x4 *>(smem) + P::SLOTS,
                    reinterpret_cast<uint32_t *>(reinterpret_cast<u32x4 *>(smem) + 2 * P::SLOTS)};
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, t = blockIdx.x;
    const Fr *const in = a.in_b[blockIdx.y];
    Fr *const out = a.out_b[blockIdx.y];
    // where the tile lives: cols — R rows of stride 2^log_inner, C adjacent columns from c0; rows — C contiguous rows of R
    // elements, row c = sub-transform (k1_0 + c, mu) (ntt_pass_cols / ntt_pass_rows have the same arithmetic)
    uint32_t c0 = 0, k1_0 = 0, mu = 0, log_m = 0;
    size_t base = 0;
    if constexpr (!ROWS) {
        const uint32_t tiles_log = a.log_inner - LC;
        c0 = (t & ((1u << tiles_log) - 1)) << LC;
        base = ((size_t)(t >> tiles_log) << (LR + a.log_inner)) + c0;
    } else {
        const uint32_t kb_log = a.log_r1 - LC;
        log_m = a.log_m1 + a.log_m2;
        k1_0 = (t & ((1u << kb_log) - 1)) << LC; mu = t >> kb_log;
    }

    FrW9 x0, x1, x2, x3;                                        // (named registers, compile-time loops: nothing of this may land in scratch)
    auto X = [&](auto K) -> FrW9 & { constexpr int k = decltype(K)::value; if constexpr (k == 0) return x0; else if constexpr (k == 1) return x1; else if constexpr (k == 2) return x2; else return x3; };
    {                                                           // round 0 reads HBM: LDS row i is input row bitrev(i)
        uint32_t row0, col;
        P::template locate<0>(wave, lane, 0, row0, col);
        auto addr = [&](uint32_t k) __attribute__((always_inline)) -> size_t {
            const uint32_t n = brev(row0 + k, LR);              // (round 0 starts at stage 0: the four rows are row0 + k)
            if constexpr (!ROWS) return base + ((size_t)n << a.log_inner) + col;
            else return (((((size_t)(k1_0 + col)) << log_m) + mu) << LR) + n;
        };
        const size_t g0 = addr(0), g1 = addr(1), g2 = addr(2), g3 = addr(3);
        // (zero-padded input, the natural-order LDE: indices >= nonzero are zeros that are neither stored nor read; a zero stays a zero
        //  through the input scaling below)
        const size_t live = a.nonzero ? (size_t)a.nonzero : ~(size_t)0;
        x0 = g0 < live ? ldw(in + g0) : w_zero<FrW>(); x1 = g1 < live ? ldw(in + g1) : w_zero<FrW>();
        x2 = g2 < live ? ldw(in + g2) : w_zero<FrW>(); x3 = g3 < live ? ldw(in + g3) : w_zero<FrW>();
        const Fr *const pre_direct = a.pre_direct_b[blockIdx.y];
        const PowTable pre = a.pre_b[blockIdx.y];
        if (pre_direct) {
            FrW9 y0, y1;
            mulw2(x0, ldw(pre_direct + g0), x1, ldw(pre_direct + g1), y0, y1); x0 = y0; x1 = y1;
            mulw2(x2, ldw(pre_direct + g2), x3, ldw(pre_direct + g3), y0, y1); x2 = y0; x3 = y1;
        } else if (pre.lo) {
            auto pair = [&](FrW9 &u, FrW9 &v, uint32_t e0, uint32_t e1) __attribute__((always_inline)) {
                FrW9 p0, p1, y0, y1;
                mulw2(ldw(pre.lo + (e0 & (POW_TAB - 1))), ldw(pre.hi + (e0 >> POW_SPLIT)), ldw(pre.lo + (e1 & (POW_TAB - 1))), ldw(pre.hi + (e1 >> POW_SPLIT)), p0, p1);
                mulw2(u, p0, v, p1, y0, y1); u = y0; v = y1;
            };
            pair(x0, x1, (uint32_t)g0, (uint32_t)g1);
            pair(x2, x3, (uint32_t)g2, (uint32_t)g3);
        }
    }
    static_for<P::NR>([&](auto RR) {
        constexpr int r = decltype(RR)::value;
        uint32_t row0, col;
        P::template locate<r>(wave, lane, 0, row0, col);
        const Tw3<FrW> t1 = wave_round_t1<LR, P::rs(r), P::rn(r)>(a.tw, row0);       // in flight across the exchange
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (r > 0) {                                  // exchange: what round r-1 left in registers -> the layout round r reads
            constexpr bool handover = P::pb(r - 1) != P::pb(r);
            uint32_t prow, pcol;
            P::template locate<r - 1>(wave, lane, 0, prow, pcol);
            static_for<4>([&](auto K) { constexpr uint32_t k = decltype(K)::value; L.put(P::template slot<r>(prow + (k << P::rs(r - 1)), pcol), X(K)); });
            if constexpr (handover) __syncthreads(); else wave_lds_fence();
            static_for<4>([&](auto K) { constexpr uint32_t k = decltype(K)::value; X(K) = L.get(P::template own_slot<r>(wave, lane, k)); });
        }
        // the round after this one hands over to phase B, i.e. writes into the other waves' regions: they must all have
        // finished reading theirs (their gets of THIS round) first
        if constexpr (r > 0 && r + 1 < P::NR && P::pb(r) != P::pb(r + 1)) __syncthreads();
        wave_round<LR, P::rs(r), P::rn(r)>(x0, x1, x2, x3, a.tw, row0, t1);
    });
    {                                                           // the last round's rows row0 + k * R/4 go straight out
        uint32_t row0, col;
        P::template locate<P::NR - 1>(wave, lane, 0, row0, col);
        if constexpr (!ROWS) {
            const uint32_t eshift = MAX_LOG_N - (LR + a.log_inner);
            auto off = [&](uint32_t k) __attribute__((always_inline)) -> size_t { return ((size_t)(row0 + (k << (LR - 2))) << a.log_inner) + c0 + col; };
            auto twd = [&](uint32_t k, size_t o) __attribute__((always_inline)) -> FrW9 {
                const uint32_t e = ((row0 + (k << (LR - 2))) * (c0 + col)) << eshift;
                return a.tw_direct ? ldw(a.tw_direct + o) : eshift >= POW_SPLIT ? ldw(a.tw.hi + (e >> POW_SPLIT)) : pow2l_w(a.tw, e);
            };
            Fr *const ob = out + (base - c0);
            auto finish = [&](const FrW9 &u, const FrW9 &v, uint32_t k) __attribute__((always_inline)) {      // (two twiddles at a time: four would not fit 128 registers)
                const size_t o0 = off(k), o1 = off(k + 1);
                const FrW9 t0 = twd(k, o0), t1 = twd(k + 1, o1);
                FrW9 y0, y1;
                mulw2(u, t0, v, t1, y0, y1);                    // < 1.1p: fits 256 bits, stays lazy
                store_fp(ob + o0, pack<FrParams>(y0)); store_fp(ob + o1, pack<FrParams>(y1));
            };
            finish(x0, x1, 0);
            finish(x2, x3, 2);
        } else {
            const uint32_t k2 = mu >> a.log_m2, k3 = mu & ((1u << a.log_m2) - 1);
            const size_t drev = (size_t)k2 + ((size_t)k3 << a.log_m1);
            const Fr *const post_direct = a.post_direct_b[blockIdx.y];
            const PowTable post = a.post_b[blockIdx.y];
            const FrW9 last = unpack<FrW>(a.scale);             // 1/n on inverse transforms
            auto finish = [&](FrW9 v0, FrW9 v1, uint32_t k) __attribute__((always_inline)) {
                const size_t o0 = (size_t)(k1_0 + col) + (drev << a.log_r1) + ((size_t)(row0 + (k << (LR - 2))) << (a.log_n - LR));
                const size_t o1 = (size_t)(k1_0 + col) + (drev << a.log_r1) + ((size_t)(row0 + ((k + 1) << (LR - 2))) << (a.log_n - LR));
                FrW9 y0, y1;
                if (post_direct) { mulw2(v0, ldw(post_direct + o0), v1, ldw(post_direct + o1), y0, y1); v0 = y0; v1 = y1; }
                else if (post.lo) {
                    const uint32_t e0 = (uint32_t)o0, e1 = (uint32_t)o1;
                    FrW9 p0, p1;
                    mulw2(ldw(post.lo + (e0 & (POW_TAB - 1))), ldw(post.hi + (e0 >> POW_SPLIT)), ldw(post.lo + (e1 & (POW_TAB - 1))), ldw(post.hi + (e1 >> POW_SPLIT)), p0, p1);
                    mulw2(v0, p0, v1, p1, y0, y1); v0 = y0; v1 = y1;
                }
                if (a.has_scale) { mulw2(v0, last, v1, last, y0, y1); v0 = csub_p(y0); v1 = csub_p(y1); }
                else { v0 = reduce_small(v0); v1 = reduce_small(v1); }       // canonical output (values here are < 24p)
                store_fp(out + o0, pack<FrParams>(v0)); store_fp(out + o1, pack<FrParams>(v1));
            };
            finish(x0, x1, 0);
            finish(x2, x3, 2);
        }
    }
}

// ------------------------------------------------------------------------------- tables
// w_domain: entries 32 * base^e, i.e. base^e * 2^261 — what the 29-bit layer's products take as a constant (poly.hip)
__global__ void fill_pow_table(Fr *lo, Fr *hi, Fr base, int w_domain = 0) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * POW_TAB) return;
    Fr v = i < POW_TAB ? pow_u64(base, i) : pow_u64(base, (uint64_t)(i - POW_TAB) << POW_SPLIT);
    if (w_domain) v = mul(v, from_u64<FrParams>(32));
    store_fp(i < POW_TAB ? lo + i : hi + (i - POW_TAB), v);
}

// canonical omega_{2^28} — 7^((r-1)/2^28)
static Fr root28() {
    Fr c;
    const uint32_t w[8] = {0x60c37c9cu, 0xd34f1ed9u, 0xd39329c8u, 0x3215cf6du, 0x3dd31f74u, 0x98865ea9u, 0x166d18b7u, 0x03ddb9f5u};
    for (int i = 0; i < 8; i++) c.l[i] = w[i];
    return from_canonical(c);
}

Fr ntt_omega(uint32_t log_n) {
    Fr w = root28();
    for (uint32_t i = log_n; i < MAX_LOG_N; i++) w = sqr(w);
    return w;
}

// the same table with every entry moved into the 2^261 domain of field29_dev.h (x*2^256 -> x*2^261)
__global__ void table_to_w(Fr *out, const Fr *in, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) store_fp(out + i, pack<FrParams>(csub_p(w_from_s(unpack<FrW>(load_fp(in + i))))));
}

// the W-domain hi table once more as 9 x 29-bit limbs padded to 48 bytes per entry
__global__ void table_slice(uint32_t *out, const Fr *in_w, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FrW9 v = unpack<FrW>(load_fp(in_w + i));
    for (int k = 0; k < 9; k++) out[12 * (size_t)i + k] = v.l[k];
    for (int k = 9; k < 12; k++) out[12 * (size_t)i + k] = 0;
}

// the W-domain hi table as mul_tw3 constants: three shifted copies of 9 limbs, padded to 112 bytes (7 x 16) per entry
constexpr uint32_t TW3_WORDS = 28;
__global__ void table_tw3(uint32_t *out, const Fr *in_w, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Tw3<FrW> t = make_tw3(unpack<FrW>(load_fp(in_w + i)));             // (table_to_w leaves canonical values)
    for (int q = 0; q < 3; q++) for (int k = 0; k < 9; k++) out[TW3_WORDS * (size_t)i + 9 * q + k] = t.w[q][k];
    out[TW3_WORDS * (size_t)i + 27] = 0;
}

// allocates [lo | hi] in the external domain followed by [lo | hi] in the W domain, the sliced W-domain hi table and (tw3: the
// transform's own twiddle tables only) the same hi table as mul_tw3 constants
static int32_t make_pow_table(plk_ctx *ctx, const Fr &base, PowTable *out, PowTable *out_w, void **alloc_out, bool tw3 = false) {
    Fr *buf = nullptr;
    PLK_HIP(hipMalloc(&buf, sizeof(Fr) * 4 * POW_TAB + (size_t)48 * POW_TAB + (tw3 ? (size_t)4 * TW3_WORDS * POW_TAB : 0)));
    hipLaunchKernelGGL(fill_pow_table, dim3(2 * POW_TAB / 256), dim3(256), 0, ctx->stream, buf, buf + POW_TAB, base, 0);
    hipLaunchKernelGGL(table_to_w, dim3(2 * POW_TAB / 256), dim3(256), 0, ctx->stream, buf + 2 * POW_TAB, (const Fr *)buf, 2 * POW_TAB);
    PLK_HIP(hipGetLastError());
    out->lo = buf;
    out->hi = buf + POW_TAB;
    out_w->lo = buf + 2 * POW_TAB;
    out_w->hi = buf + 3 * POW_TAB;
    uint32_t *sliced = reinterpret_cast<uint32_t *>(buf + 4 * POW_TAB);
    hipLaunchKernelGGL(table_slice, dim3(POW_TAB / 256), dim3(256), 0, ctx->stream, sliced, (const Fr *)(buf + 3 * POW_TAB), POW_TAB);
    PLK_HIP(hipGetLastError());
    out_w->hi_sliced = sliced;
    if (tw3) {
        uint32_t *t3 = sliced + (size_t)12 * POW_TAB;
        hipLaunchKernelGGL(table_tw3, dim3(POW_TAB / 256), dim3(256), 0, ctx->stream, t3, (const Fr *)(buf + 3 * POW_TAB), POW_TAB);
        PLK_HIP(hipGetLastError());
        out_w->hi_tw3 = t3;
    }
    if (alloc_out) *alloc_out = buf;
    return PLK_OK;
}

Fr cached_inverse(plk_ctx *ctx, const Fr &g) {
    std::vector<uint32_t> key(g.l, g.l + 8);
    auto it = ctx->inv_cache.find(key);
    if (it != ctx->inv_cache.end()) return it->second;
    Fr gi = inv(g);
    ctx->inv_cache[key] = gi;
    return gi;
}

// four tables in ONE launch (round 4 of the prover needs z, 1/z, z*omega, 1/(z*omega): four launches of 42 us each were pure latency).
// Round 6: every thread used to compute its entry base^e with its own square-and-multiply — a dependent chain of up to 28 products, 68 us of a
// proof at EVERY domain size (4 % of a 2^12 proof).  The chain of squarings base^(2^j), j < 28, is the same for all threads of a table: the
// host computes it (27 squarings of 30 ns) and hands it over as a kernel argument; an entry is then the product of the 14 squares its
// exponent selects — a fixed tree of 13 independent-by-level products (a factor whose bit is clear is the Montgomery one: no divergence).
// The products run on the 29-bit layer in lockstep pairs (mulw2).  Every mulw divides by 2^261, i.e. leaves 2^-5 relative to the R = 2^256 form the
// squares come in; the tree has 13 products whatever the exponent, so ONE last product by `fix` = 2^(65 + 5 + 261) brings the result to base^e * 2^261
// (the W domain the tables are read in) — 14 products of ~0.5 us per pair instead of up to 28 dependent ones of the 32-bit layer: 68 -> ~15 us.
struct FourSquares { Fr sq[4][28]; Fr *buf[4]; Fr fix; };
__global__ void fill_pow_tables4(FourSquares a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (i >= 2 * POW_TAB) return;
    Fr *lo = a.buf[k], *hi = lo + POW_TAB;
    const bool is_hi = i >= POW_TAB;
    const uint32_t e = is_hi ? i - POW_TAB : i;                    // 14 bits; the hi table's exponents are e << 14: squares 14..27
    FrW9 f[14];
    const FrW9 one = unpack<FrW>(Fr::one());
#pragma unroll
    for (int j = 0; j < 14; j++) {
        const FrW9 q = unpack<FrW>(is_hi ? a.sq[k][14 + j] : a.sq[k][j]);
        const uint32_t on = 0u - ((e >> j) & 1u);
#pragma unroll
        for (int l = 0; l < 9; l++) f[j].l[l] = (q.l[l] & on) | (one.l[l] & ~on);
    }
    FrW9 g[7];
    mulw2(f[0], f[1], f[2], f[3], g[0], g[1]); mulw2(f[4], f[5], f[6], f[7], g[2], g[3]); mulw2(f[8], f[9], f[10], f[11], g[4], g[5]);
    g[6] = mulw(f[12], f[13]);
    FrW9 h0, h1, h2, h3;
    mulw2(g[0], g[1], g[2], g[3], h0, h1);
    mulw2(g[4], g[5], h0, h1, h2, h3);                            // h2 = g4 g5, h3 = (g0 g1)(g2 g3)
    FrW9 v = mulw(mulw(h2, g[6]), h3);
    v = csub_p(mulw(v, unpack<FrW>(a.fix)));
    store_fp(is_hi ? hi + e : lo + e, pack<FrParams>(v));
}
int32_t fill_pow_tables4_into(plk_ctx *, const Fr bases[4], Fr *const bufs[4], PowTable out[4], hipStream_t s) {
    FourSquares a;
    for (int k = 0; k < 4; k++) {
        a.buf[k] = bufs[k]; out[k].lo = bufs[k]; out[k].hi = bufs[k] + POW_TAB;
        host::HFr x; memcpy(x.l, bases[k].l, 32);                  // (same Montgomery representation on both sides: hostmath.h)
        for (int j = 0; j < 28; j++) { memcpy(a.sq[k][j].l, x.l, 32); x = x.sqr(); }
    }
    // fix = 2^331 as a plain residue: 13 products x 2^-5, the last product's own 2^-261, and the W domain's 2^5 over R = 2^256
    static const host::HFr fix = [] { host::HFr t = host::HFr::from_u64(2), r = host::HFr::one(); for (int b = 0; b < 331; b++) r = r * t; uint64_t c[4]; r.to_canonical(c); host::HFr o; memcpy(o.l, c, 32); return o; }();
    memcpy(a.fix.l, fix.l, 32);
    hipLaunchKernelGGL(fill_pow_tables4, dim3(2 * POW_TAB / 256, 4), dim3(256), 0, s, a);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

int32_t fill_pow_table_into(plk_ctx *ctx, const Fr &base, Fr *buf, PowTable *out, hipStream_t s) {      // W-domain table
    hipLaunchKernelGGL(fill_pow_table, dim3(2 * POW_TAB / 256), dim3(256), 0, s, buf, buf + POW_TAB, base, 1);
    PLK_HIP(hipGetLastError());
    out->lo = buf;
    out->hi = buf + POW_TAB;
    return PLK_OK;
}

int32_t ntt_init_tables(plk_ctx *ctx) {
    if (ctx->tw_fwd.lo) return PLK_OK;
    Fr half = inv(from_u64<FrParams>(2));
    ctx->n_inv[0] = Fr::one();
    for (uint32_t i = 1; i <= MAX_LOG_N; i++) ctx->n_inv[i] = mul(ctx->n_inv[i - 1], half);
    Fr w = root28();
    void *a = nullptr, *b = nullptr;
    PLK_TRY(make_pow_table(ctx, w, &ctx->tw_fwd, &ctx->tw_fwd_w, &a, true));
    PLK_TRY(make_pow_table(ctx, inv(w), &ctx->tw_inv, &ctx->tw_inv_w, &b, true));
    for (uint32_t i = 0; i <= MAX_LOG_N; i++) ctx->n_inv_w[i] = mul(ctx->n_inv[i], from_u64<FrParams>(32));   // x*2^256 -> x*2^261
    ctx->coset_allocs.push_back(a);
    ctx->coset_allocs.push_back(b);
    // the tables are filled on the context's stream but read by transforms on ANY stream (plk_ntt_dev takes the caller's):
    // they must be complete before the first of those is enqueued.  Once per context.
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return PLK_OK;
}

// returns the power table of g in the W domain (what the NTT passes consume)
int32_t ntt_coset_table(plk_ctx *ctx, const Fr &g, PowTable *out) {
    std::vector<uint32_t> key(g.l, g.l + 8);
    auto it = ctx->coset_tabs.find(key);
    if (it != ctx->coset_tabs.end()) { *out = it->second; return PLK_OK; }
    void *a = nullptr;
    PowTable ext;
    PLK_TRY(make_pow_table(ctx, g, &ext, out, &a));
    ctx->coset_allocs.push_back(a);
    ctx->coset_tabs[key] = *out;
    PLK_HIP(hipStreamSynchronize(ctx->stream));               // filled on the context's stream, used on any (see ntt_init_tables); once per shift
    return PLK_OK;
}

// ------------------------------------------------------------------------------- inter-pass twiddle tables
// A pass over a sub-transform of more than 2^14 points used to compose every inter-pass twiddle from the two-level
// table (one product per element, a tenth of the transform's arithmetic at 2^20).  The passes are bound by
// instruction issue and use 6 % of the HBM bandwidth, so the twiddles of such a pass are kept as a table of their
// own instead — 32 B more read per element and pass — with 1/n folded into the first table of an inverse transform
// (the last pass then only canonicalises).  Built at the first use of a (direction, digit plan) pair, kept for
// the life of the context; 32 MiB per table at 2^20, capped at 1 GiB (2^25 entries): beyond that, and with
// PLK_NTT_DIRECT=0, the passes compose as before.
__global__ void ntt_fill_direct(Fr *out, PowTable tw, uint32_t log_r, uint32_t log_inner, Fr scale_w, uint32_t has_scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> (log_r + log_inner)) return;
    const uint32_t k = (uint32_t)(i >> log_inner), col = (uint32_t)i & ((1u << log_inner) - 1);
    const uint32_t e = (k * col) << (MAX_LOG_N - (log_r + log_inner));
    FrW9 v = pow2l_w(tw, e);
    if (has_scale) v = mulw(csub_p(v), unpack<FrW>(scale_w));
    store_fp(out + i, pack<FrParams>(csub_p(v)));
}

constexpr uint32_t NTT_DIRECT_MAX_LOG = 25;

static bool ntt_direct_enabled() {
    static const int on = [] { const char *e = getenv("PLK_NTT_DIRECT"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0;
}

// returns nullptr (and PLK_OK) when the pass should compose its twiddles
static int32_t ntt_direct_table(plk_ctx *ctx, bool inverse, uint32_t log_r, uint32_t log_inner, bool scaled, uint32_t log_n,
                                hipStream_t stream, const Fr **out) {
    *out = nullptr;
    const uint32_t bits = log_r + log_inner;
    if (!ntt_direct_enabled() || MAX_LOG_N - bits >= POW_SPLIT || bits > NTT_DIRECT_MAX_LOG) return PLK_OK;
    const uint32_t key = (inverse ? 1u << 31 : 0) | (scaled ? log_n << 16 : 0) | log_r << 8 | log_inner;
    ctx->ntt_direct_clock++;
    auto it = ctx->ntt_direct.find(key);
    if (it != ctx->ntt_direct.end()) { ctx->ntt_direct_used[key] = ctx->ntt_direct_clock; *out = static_cast<const Fr *>(it->second); return PLK_OK; }
    // A long-lived context that proves many domain sizes would otherwise collect tables for ever (up to 1 GiB each).  When the next
    // one would take the total past the cap, the tables that have not been asked for during the last NTT_DIRECT_KEEP requests (about
    // two proofs' worth) go, oldest first, until it fits — after the device has drained, since passes in flight on any stream may
    // still be reading them.  Tables of the plan in use are never dropped: if they alone fill the cap (a small
    // PLK_NTT_DIRECT_CAP_MB, or one proof whose shapes exceed it) the new pass composes its twiddles instead — round 3 dropped
    // everything, which made such a proof rebuild its tables, and stall every stream, on every pass.
    static const size_t cap = [] { const char *e = getenv("PLK_NTT_DIRECT_CAP_MB"); size_t mb = e ? strtoull(e, nullptr, 10) : 0; return (mb ? mb : 6144) << 20; }();
    const size_t want = sizeof(Fr) << bits;
    if (ctx->ntt_direct_bytes + want > cap) {
        constexpr uint64_t NTT_DIRECT_KEEP = 96;
        std::vector<std::pair<uint64_t, uint32_t>> idle;             // (last use, key) of the evictable tables
        for (auto &kv : ctx->ntt_direct) { const uint64_t u = ctx->ntt_direct_used[kv.first]; if (u + NTT_DIRECT_KEEP < ctx->ntt_direct_clock) idle.emplace_back(u, kv.first); }
        std::sort(idle.begin(), idle.end());
        size_t freed = 0;
        for (auto &e : idle) { if (ctx->ntt_direct_bytes - freed + want <= cap) break; freed += ctx->ntt_direct_size[e.second]; }
        if (ctx->ntt_direct_bytes - freed + want > cap) return PLK_OK;            // the tables in use fill the cap: compose
        PLK_HIP(hipDeviceSynchronize());
        for (auto &e : idle) {
            if (ctx->ntt_direct_bytes + want <= cap) break;
            (void)hipFree(ctx->ntt_direct[e.second]);
            ctx->ntt_direct_bytes -= ctx->ntt_direct_size[e.second];
            ctx->ntt_direct.erase(e.second); ctx->ntt_direct_used.erase(e.second); ctx->ntt_direct_size.erase(e.second);
        }
    }
    Fr *buf = nullptr;
    if (hipMalloc(&buf, want) != hipSuccess) { (void)hipGetLastError(); return PLK_OK; }   // no room: compose
    const PowTable &tw = inverse ? ctx->tw_inv_w : ctx->tw_fwd_w;
    hipLaunchKernelGGL(ntt_fill_direct, dim3((uint32_t)((((size_t)1 << bits) + 255) / 256)), dim3(256), 0, stream, buf, tw, log_r, log_inner,
                       scaled ? ctx->n_inv_w[log_n] : Fr::zero(), scaled ? 1u : 0u);
    PLK_HIP(hipGetLastError());
    PLK_HIP(hipStreamSynchronize(stream));                 // other streams may use the table right after this call
    ctx->ntt_direct[key] = buf;
    ctx->ntt_direct_used[key] = ctx->ntt_direct_clock;
    ctx->ntt_direct_size[key] = want;
    ctx->ntt_direct_bytes += want;
    *out = buf;
    return PLK_OK;
}

// ------------------------------------------------------------------------------- driver
static std::atomic<bool> g_attr_set{false};                    // (several contexts may transform from several host threads)

// the wave-owned passes (ntt_pass_w) take every full 2048-element tile of 7..10 row bits; PLK_NTT_WAVE=0 keeps the
// barrier-per-round kernels for everything (A/B knob)
constexpr size_t NTT_W_LDS = (size_t)36 * NTT_W_SLOTS, NTT_W_LDS_BIG = (size_t)36 * NTT_W_SLOTS_BIG;
static bool ntt_wave_enabled() {
    static const int on = [] { const char *e = getenv("PLK_NTT_WAVE"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0;
}
// the 4096-element tile (2^21 / 2^22-point transforms in two passes); PLK_NTT_BIG_TILE=0: three passes of 2048-element tiles as before (A/B knob)
static bool ntt_big_tile_enabled() {
    static const int on = [] { const char *e = getenv("PLK_NTT_BIG_TILE"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0 && ntt_wave_enabled();
}
static bool ntt_wave_shape(const NttPassArgs &a) {
    // (zero-padded inputs are taken too — `quarter`, the old kernels' copy-instead-of-butterfly shortcut for them, is simply not used;
    //  their per-element coset tables are never combined with padding)
    if (!ntt_wave_enabled() || (a.nonzero && a.pre_direct_b[0])) return false;
    if (a.log_r + a.log_c == LOG_TILE) return a.log_r >= 7 && a.log_r <= 10;
    return a.log_r + a.log_c == (uint32_t)NTT_LOG_TILE_BIG && a.log_r >= 10 && a.log_r <= 11 && ntt_big_tile_enabled();
}
template <bool ROWS>
static void ntt_launch_w(const NttPassArgs &a, dim3 grid, hipStream_t stream) {
    if (a.log_r + a.log_c == (uint32_t)NTT_LOG_TILE_BIG) {
        if (a.log_r == 10) hipLaunchKernelGGL((ntt_pass_w<10, ROWS, NTT_LOG_TILE_BIG>), grid, dim3(2 * NTT_THREADS), NTT_W_LDS_BIG, stream, a);
        else hipLaunchKernelGGL((ntt_pass_w<11, ROWS, NTT_LOG_TILE_BIG>), grid, dim3(2 * NTT_THREADS), NTT_W_LDS_BIG, stream, a);
        return;
    }
    switch (a.log_r) {
        case 7: hipLaunchKernelGGL((ntt_pass_w<7, ROWS>), grid, dim3(NTT_THREADS), NTT_W_LDS, stream, a); break;
        case 8: hipLaunchKernelGGL((ntt_pass_w<8, ROWS>), grid, dim3(NTT_THREADS), NTT_W_LDS, stream, a); break;
        case 9: hipLaunchKernelGGL((ntt_pass_w<9, ROWS>), grid, dim3(NTT_THREADS), NTT_W_LDS, stream, a); break;
        default: hipLaunchKernelGGL((ntt_pass_w<10, ROWS>), grid, dim3(NTT_THREADS), NTT_W_LDS, stream, a); break;
    }
}
template <int LR> static hipError_t ntt_w_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ntt_pass_w<LR, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_W_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(ntt_pass_w<LR, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_W_LDS);
}
template <int LR> static hipError_t ntt_w_attr_big() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ntt_pass_w<LR, false, NTT_LOG_TILE_BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_W_LDS_BIG);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(ntt_pass_w<LR, true, NTT_LOG_TILE_BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_W_LDS_BIG);
}

// digits of the mixed-radix plan: up to 10 bits per pass
static void digit_plan(uint32_t log_n, uint32_t d[4], uint32_t *passes) {
    d[0] = d[1] = d[2] = d[3] = 0;
    uint32_t p = 1;
    // one pass — ONE workgroup holds the whole transform in LDS — up to 2^11 points.  2^12 points in one workgroup (147 KB of the CU's 160 KB, the kernel
    // handles it) was tried in round 6 and is SLOWER than two passes of two workgroups: 62 against 37 us per transform, same box (six rounds of two
    // radix-4 groups per thread on one CU against two launches of three rounds on two)
    if (log_n <= LOG_SINGLE) d[0] = log_n;
    else {
        p = (log_n + 9) / 10;                              // up to 10 bits per pass: 2^20 = 10 + 10 (two passes)
        for (uint32_t i = 0; i < p; i++) d[i] = log_n / p + (i < log_n % p ? 1 : 0);
        // 11-bit digits on the 4096-element tile: 2^21 = 11 + 10 in two passes instead of three (0.207 -> 0.194 ms, same box).  2^22 = 11 + 11 is NOT
        // taken: both passes then run on 64-byte row segments (C = 2) and the transform is 12 % SLOWER than three passes of the 2048-element tile
        // (0.375 -> 0.421 ms; profiles/r06_ntt_big_tile_ab.txt) — PLK_NTT_BIG_TILE=2 forces it for that measurement
        static const bool force22 = [] { const char *e = getenv("PLK_NTT_BIG_TILE"); return e && e[0] == '2'; }();
        if ((log_n == 21 || (log_n == 22 && force22)) && ntt_big_tile_enabled()) {
            d[0] = 11; d[1] = log_n - 11; d[2] = 0; *passes = 2;
            return;
        }
        if (p == 3 && log_n <= 23) {                       // leave 14 bits to passes 2 and 3: their inter-pass twiddles
            d[0] = log_n - 14; d[1] = 7; d[2] = 7;         // then come straight out of the hi table (ntt_pass_cols)
        }
    }
    *passes = p;
}

// src: where the first pass reads (data itself for an in-place transform); nonzero: see NttPassArgs.
// `count` transforms of the same shape share every launch (grid.y): a 2^20-point pass is ONE round of 512 tiles on the
// chip's 512 workgroup slots, i.e. its time is the latency of a tile's load -> LDS stages -> store chain; four vectors
// per launch run at the throughput rate instead (the four wire polynomials of round 1, their four extensions).
// `lane` selects the ping-pong scratch: transforms enqueued on different streams at the same time must not share it
// (lane 1 = the prover's background stream).
static int32_t ntt_run(plk_ctx *ctx, const Fr *const *src, uint64_t nonzero, Fr *const *data, uint32_t count, uint32_t log_n, bool inverse,
                       const Fr *coset, hipStream_t stream, uint32_t lane, const PowTable *pre_each = nullptr, const PowTable *post_each = nullptr,
                       const Fr *const *pre_direct = nullptr, const Fr *const *post_direct = nullptr);

int32_t ntt_dev(plk_ctx *ctx, Fr *data, uint32_t log_n, bool inverse, const Fr *coset, hipStream_t stream) {
    const Fr *src = data;
    return ntt_run(ctx, &src, 0, &data, 1, log_n, inverse, coset, stream, 0);
}

int32_t ntt_batch_dev(plk_ctx *ctx, Fr *const *data, uint32_t count, uint32_t log_n, bool inverse, const Fr *coset, hipStream_t stream, uint32_t lane) {
    for (uint32_t done = 0; done < count;) {
        const uint32_t b = count - done > NTT_MAX_BATCH ? NTT_MAX_BATCH : count - done;
        const Fr *src[NTT_MAX_BATCH];
        for (uint32_t k = 0; k < b; k++) src[k] = data[done + k];
        PLK_TRY(ntt_run(ctx, src, 0, data + done, b, log_n, inverse, coset, stream, lane));
        done += b;
    }
    return PLK_OK;
}

// pre_each / post_each: one input- / output-scaling table per transform of the batch (the four cosets of lde4cm_batch_dev and
// of icoset4cm_dev) instead of `coset`
static int32_t ntt_run(plk_ctx *ctx, const Fr *const *src, uint64_t nonzero, Fr *const *data, uint32_t count, uint32_t log_n, bool inverse,
                       const Fr *coset, hipStream_t stream, uint32_t lane, const PowTable *pre_each, const PowTable *post_each,
                       const Fr *const *pre_direct, const Fr *const *post_direct) {
    if (!data || !src || count == 0 || count > NTT_MAX_BATCH || lane >= 2) { set_error("ntt: bad argument"); return PLK_ERR_ARG; }
    for (uint32_t b = 0; b < count; b++) if (!data[b] || !src[b]) { set_error("ntt: null data"); return PLK_ERR_ARG; }
    if (log_n > MAX_LOG_N) { set_error("ntt: log_n exceeds the 2-adicity of Fr (28)"); return PLK_ERR_SIZE; }
    if (log_n == 0) return PLK_OK;
    PLK_TRY(ntt_init_tables(ctx));
    if (!g_attr_set) {
        PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ntt_pass_rows), hipFuncAttributeMaxDynamicSharedMemorySize, 36 << LOG_SINGLE));
        PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ntt_pass_cols), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        PLK_HIP(ntt_w_attr<7>()); PLK_HIP(ntt_w_attr<8>()); PLK_HIP(ntt_w_attr<9>()); PLK_HIP(ntt_w_attr<10>());
        PLK_HIP(ntt_w_attr_big<10>()); PLK_HIP(ntt_w_attr_big<11>());
        g_attr_set = true;
    }
    PowTable pre{}, post{};
    if (coset) {
        if (!inverse) PLK_TRY(ntt_coset_table(ctx, *coset, &pre));
        else PLK_TRY(ntt_coset_table(ctx, cached_inverse(ctx, *coset), &post));
    }
    uint32_t d[4], p;
    digit_plan(log_n, d, &p);
    const uint32_t log_tile = (p == 2 && d[0] == 11) ? (uint32_t)NTT_LOG_TILE_BIG : (uint32_t)LOG_TILE;     // 11-bit digits: the 4096-element tile
    const size_t n = (size_t)1 << log_n;
    if (p > 1) PLK_TRY(ctx->ntt_scratch[lane].reserve((size_t)count * n * sizeof(Fr)));
    Fr *const scratch = ctx->ntt_scratch[lane].as<Fr>();

    NttPassArgs a{};
    a.log_n = log_n;
    a.tw = inverse ? ctx->tw_inv_w : ctx->tw_fwd_w;
    // ping-pong: pass 1 data -> scratch, middle passes in place in scratch, last pass scratch -> data
    uint32_t rem = log_n;
    bool scale_folded = false;
    for (uint32_t i = 0; i + 1 < p; i++) {
        rem -= d[i];
        for (uint32_t b = 0; b < count; b++) { a.in_b[b] = (i == 0) ? src[b] : scratch + (size_t)b * n; a.out_b[b] = scratch + (size_t)b * n; }
        a.nonzero = (i == 0) ? (uint32_t)nonzero : 0;
        a.log_r = d[i]; a.log_inner = rem;
        a.log_c = (log_tile - d[i]) < rem ? (log_tile - d[i]) : rem;
        for (uint32_t b = 0; b < count; b++) a.pre_b[b] = (i == 0) ? (pre_each ? pre_each[b] : pre) : PowTable{};
        for (uint32_t b = 0; b < count; b++) a.post_b[b] = PowTable{};
        for (uint32_t b = 0; b < count; b++) { a.pre_direct_b[b] = (i == 0 && pre_direct) ? pre_direct[b] : nullptr; a.post_direct_b[b] = nullptr; }
        a.has_scale = 0;
        PLK_TRY(ntt_direct_table(ctx, inverse, d[i], rem, inverse && i == 0, log_n, stream, &a.tw_direct));
        if (a.tw_direct && inverse && i == 0) scale_folded = true;
        a.quarter = (i == 0 && nonzero && nonzero == (n >> 2) && !(d[0] & 1) && d[0] >= 2) ? 1 : 0;
        uint32_t tiles = (uint32_t)(n >> (a.log_r + a.log_c));
        size_t lds = (size_t)36 << (a.log_r + a.log_c);
        if (ntt_wave_shape(a)) ntt_launch_w<false>(a, dim3(tiles, count), stream);
        else hipLaunchKernelGGL(ntt_pass_cols, dim3(tiles, count), dim3(NTT_THREADS), lds, stream, a);
    }
    {
        // (a single pass is one workgroup per transform that has read ALL of its input into LDS — a barrier — before its first store: in place is safe,
        //  no detour through the scratch buffer and no copy launch behind it)
        for (uint32_t b = 0; b < count; b++) { a.in_b[b] = (p == 1) ? src[b] : scratch + (size_t)b * n; a.out_b[b] = data[b]; }
        a.nonzero = (p == 1) ? (uint32_t)nonzero : 0;
        a.log_r = d[p - 1];
        if (p == 1) { a.log_r1 = 0; a.log_m1 = a.log_m2 = 0; a.log_c = 0; }
        else {
            a.log_r1 = d[0];
            a.log_m1 = p >= 3 ? d[1] : 0;
            a.log_m2 = p >= 4 ? d[2] : 0;
            a.log_c = (log_tile - a.log_r) < d[0] ? (log_tile - a.log_r) : d[0];
        }
        for (uint32_t b = 0; b < count; b++) a.pre_b[b] = (p == 1) ? (pre_each ? pre_each[b] : pre) : PowTable{};
        for (uint32_t b = 0; b < count; b++) a.post_b[b] = post_each ? post_each[b] : post;
        for (uint32_t b = 0; b < count; b++) { a.pre_direct_b[b] = (p == 1 && pre_direct) ? pre_direct[b] : nullptr; a.post_direct_b[b] = post_direct ? post_direct[b] : nullptr; }
        a.tw_direct = nullptr; a.quarter = 0;
        a.has_scale = (inverse && !scale_folded) ? 1 : 0;
        if (a.has_scale) a.scale = ctx->n_inv_w[log_n];
        uint32_t tiles = (uint32_t)(n >> (a.log_r + a.log_c));
        size_t lds = (size_t)36 << (a.log_r + a.log_c);
        if (p > 1 && ntt_wave_shape(a)) ntt_launch_w<true>(a, dim3(tiles, count), stream);
        else hipLaunchKernelGGL(ntt_pass_rows, dim3(tiles, count), dim3(NTT_THREADS), lds, stream, a);
    }
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

// ------------------------------------------------------------------------------- coset-shift tables of the prover's extensions
// lde4cm_batch_dev scales the input of coset k by g_k^i (g_k = 7 * omega_4n^k), icoset4cm_dev the output by g_k^-j.  From the
// two-level power table that is two loads and TWO products per element (the composition lo * hi is a product itself) — a fifth
// of the arithmetic of an extension, whose passes are bound by instruction issue at 5 % of the HBM bandwidth.  For these two
// callers the powers are kept as tables of their own: 4 x n entries per (log_n, direction), W domain, built at first use.
// 128 MiB at 2^20; above 2^24 (2 GiB) the passes compose as before.  One entry per direction is kept (the domain in use).
__global__ void ntt_fill_coset_direct(Fr *out, PowTable t0, PowTable t1, PowTable t2, PowTable t3, uint32_t log_n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> (log_n + 2)) return;
    const uint32_t k = (uint32_t)(i >> log_n), e = (uint32_t)i & ((1u << log_n) - 1);
    const PowTable &t = k == 0 ? t0 : (k == 1 ? t1 : (k == 2 ? t2 : t3));
    store_fp(out + i, pack<FrParams>(csub_p(pow2l_w(t, e))));
}
static int32_t coset_direct_tables(plk_ctx *ctx, uint32_t log_n, bool inverse, const PowTable tabs[4], hipStream_t stream, const Fr *out[4]) {
    for (int k = 0; k < 4; k++) out[k] = nullptr;
    static const bool on = [] { const char *e = getenv("PLK_NTT_COSET_DIRECT"); return !(e && e[0] == '0'); }();       // A/B knob
    if (!on || !ntt_direct_enabled() || log_n > 24 || log_n <= POW_SPLIT) return PLK_OK;      // (<= 2^14 points: the low table alone holds the power)
    plk_ctx::CosetDirect &C = ctx->coset_direct[inverse ? 1 : 0];
    const size_t n = (size_t)1 << log_n;
    if (C.log_n != log_n || !C.buf.p) {
        if (C.buf.p) PLK_HIP(hipDeviceSynchronize());        // passes of another domain may still be reading the old tables
        C.log_n = 0;
        if (C.buf.cap < 4 * n * sizeof(Fr)) C.buf.release();
        if (C.buf.reserve(4 * n * sizeof(Fr)) != PLK_OK) { (void)hipGetLastError(); return PLK_OK; }       // no room: compose
        hipLaunchKernelGGL(ntt_fill_coset_direct, dim3((uint32_t)((4 * n + 255) / 256)), dim3(256), 0, stream, C.buf.as<Fr>(), tabs[0], tabs[1], tabs[2], tabs[3], log_n);
        PLK_HIP(hipGetLastError());
        PLK_HIP(hipStreamSynchronize(stream));               // other streams may use the tables right after this call
        C.log_n = log_n;
    }
    for (int k = 0; k < 4; k++) out[k] = C.buf.as<Fr>() + (size_t)k * n;
    return PLK_OK;
}

int32_t lde4_batch_dev(plk_ctx *ctx, const Fr *const *coeffs, uint32_t count, uint32_t log_n, Fr *const *out_4n, hipStream_t stream, uint32_t lane) {
    if (log_n + 2 > MAX_LOG_N) { set_error("lde4: 4n exceeds 2^28"); return PLK_ERR_SIZE; }
    const size_t n = (size_t)1 << log_n;
    // the zero-padded 4n-point coset transform without materialising the padding: the first pass reads the n
    // coefficients where they are and treats every index >= n as zero (no copy, no memset, no scaling of zeros)
    for (uint32_t k = 0; k < count; k++) if (coeffs[k] == out_4n[k]) { set_error("lde4: input and output must not alias"); return PLK_ERR_ARG; }
    Fr g = from_u64<FrParams>(7);
    // one launch per pass for the whole batch while its ping-pong scratch stays below 2 GiB (count x 4n x 32 B)
    const uint32_t per = log_n + 2 <= 22 ? NTT_MAX_BATCH : (log_n + 2 <= 24 ? 4 : 1);
    for (uint32_t done = 0; done < count;) {
        const uint32_t b = count - done > per ? per : count - done;
        PLK_TRY(ntt_run(ctx, coeffs + done, n, out_4n + done, b, log_n + 2, false, &g, stream, lane));
        done += b;
    }
    return PLK_OK;
}

// The inverse of lde4cm_batch_dev's first half: 4n values in coset-major order -> for every coset k the coefficients u_k of the
// degree-< n polynomial that takes them on g_k * <omega_n>, in place (u_k at data + k * n): four inverse n-point transforms with
// their output scaled by g_k^-j, one launch per pass.  poly.hip's icoset_combine turns the four u_k into the 4n coefficients.
int32_t icoset4cm_dev(plk_ctx *ctx, Fr *data_4n, uint32_t log_n, hipStream_t stream, uint32_t lane) {
    if (log_n + 2 > MAX_LOG_N) { set_error("icoset4: 4n exceeds 2^28"); return PLK_ERR_SIZE; }
    const size_t n = (size_t)1 << log_n;
    PLK_TRY(ntt_init_tables(ctx));
    PowTable post[4];
    {
        Fr g = from_u64<FrParams>(7);
        const Fr w = ntt_omega(log_n + 2);
        for (int k = 0; k < 4; k++) { PLK_TRY(ntt_coset_table(ctx, cached_inverse(ctx, g), &post[k])); g = mul(g, w); }
    }
    const Fr *src[4]; Fr *dst[4];
    for (uint32_t k = 0; k < 4; k++) { src[k] = data_4n + k * n; dst[k] = data_4n + k * n; }
    const Fr *direct[4];
    PLK_TRY(coset_direct_tables(ctx, log_n, true, post, stream, direct));
    return ntt_run(ctx, src, 0, dst, 4, log_n, true, nullptr, stream, lane, nullptr, post, nullptr, direct[0] ? direct : nullptr);
}

int32_t lde4_dev(plk_ctx *ctx, const Fr *coeffs, uint32_t log_n, Fr *out_4n, hipStream_t stream) {
    return lde4_batch_dev(ctx, &coeffs, 1, log_n, &out_4n, stream, 0);
}

// The same 4n evaluations in COSET-MAJOR order: out[k * n + r] = f(7 * omega_4n^(4 r + k)), k = 0..3 — the layout the
// prover's round 3 works in.  The coset 7 * <omega_4n> is the union of the four cosets g_k * <omega_n>, g_k = 7 * omega_4n^k,
// so the extension is four n-point coset transforms of the SAME coefficients (input scaled by g_k^i on load): 4 n log n
// butterflies instead of 4 n (log n + 2), and — what counts on this chip — two passes over n elements per coset (8 n
// element-passes at 2^20) instead of three passes over 4 n (12 n), with all cosets of up to four polynomials in one launch
// per pass.  Natural order (lde4_dev) needs the interleaving k + 4 r, i.e. 32-byte stores at a 128-byte stride; the
// quotient kernel is point-wise and does not care, so it reads this layout and interleaves only its one output vector.
int32_t lde4cm_batch_dev(plk_ctx *ctx, const Fr *const *coeffs, uint32_t count, uint32_t log_n, Fr *const *out_4n, hipStream_t stream, uint32_t lane) {
    if (log_n + 2 > MAX_LOG_N) { set_error("lde4: 4n exceeds 2^28"); return PLK_ERR_SIZE; }
    const size_t n = (size_t)1 << log_n;
    for (uint32_t k = 0; k < count; k++) if (coeffs[k] == out_4n[k]) { set_error("lde4: input and output must not alias"); return PLK_ERR_ARG; }
    PLK_TRY(ntt_init_tables(ctx));
    PowTable pre[4];
    {
        Fr g = from_u64<FrParams>(7);
        const Fr w = ntt_omega(log_n + 2);
        for (int k = 0; k < 4; k++) { PLK_TRY(ntt_coset_table(ctx, g, &pre[k])); g = mul(g, w); }
    }
    const Fr *direct[4];
    PLK_TRY(coset_direct_tables(ctx, log_n, false, pre, stream, direct));
    // polynomials per launch: 4 transforms each, at most NTT_MAX_BATCH per launch and 2 GiB of ping-pong scratch
    uint32_t per = NTT_MAX_BATCH / 4;
    while (per > 1 && (size_t)per * 4 * n * sizeof(Fr) > ((size_t)2 << 30)) per--;
    for (uint32_t done = 0; done < count;) {
        const uint32_t b = count - done > per ? per : count - done;
        const Fr *src[NTT_MAX_BATCH]; Fr *dst[NTT_MAX_BATCH]; PowTable pe[NTT_MAX_BATCH]; const Fr *pd[NTT_MAX_BATCH];
        for (uint32_t q = 0; q < b; q++)
            for (uint32_t k = 0; k < 4; k++) { src[4 * q + k] = coeffs[done + q]; dst[4 * q + k] = out_4n[done + q] + k * n; pe[4 * q + k] = pre[k]; pd[4 * q + k] = direct[k]; }
        PLK_TRY(ntt_run(ctx, src, 0, dst, 4 * b, log_n, false, nullptr, stream, lane, pe, nullptr, direct[0] ? pd : nullptr));
        done += b;
    }
    return PLK_OK;
}

}  // namespace plk
