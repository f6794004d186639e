// Short commitments (round 6): a KZG commitment of <= 2^15 terms in three short launches.
//
// The pipeline of msm.hip is tuned for 2^20 terms: 2^16 buckets, eight launches, and a bucket reduction that is a chain of ~90
// dependent full additions — at 2^12..2^14 terms a commitment costs 0.46-0.63 ms of which 0.35 ms is that chain, and the
// reference's own CI circuit (test/test_poseidon_plonk.sh: a 2^12 domain) spends 80 % of a proof there.  Same group element as
// commit_using_monomials -> dense_multiexp (src/plonk.rs:152-159), other schedule:
//   * the fixed-base table of msm.hip stays (copy w holds 2^(17w) * P_i), so all 15 signed 17-bit windows of a commitment share
//     one bucket space; a digit's magnitude m <= 2^16 is split m = 256 * hi + lo and the entry is dropped into TWO small bucket
//     sets — lo in [1, 255], hi in [1, 256] — : 30 mixed additions per term instead of 15, into 511 buckets instead of 65536;
//   * msm_small_accumulate: a workgroup of 256 lanes takes ch = 64 / 128 / 256 terms and 64 of the 512 buckets, FOUR lanes per bucket.
//     The first waves recode the scalars and append every entry to its bucket's list (LDS atomics), then lane j of a bucket adds up
//     entries j, j + 4, .. of the list (a constant column — ch equal scalars in one bucket — is ch / 4 additions per lane);
//   * msm_small_fold: the 4 n / ch partial sums of a bucket (one per lane that owned it), a tree of four-lane full additions;
//   * msm_small_planes: sum_j j * B_j = sum_b 2^b * (sum of the B_j with bit b of j set): seventeen plain tree sums over <= 128
//     buckets, side by side; the seventeen points go to the host, whose Horner (16 doublings + 16 additions) takes ~6 us.
// The longest dependent chain is ~6 mixed + 17 four-lane full additions of 2.6 us (3 in the accumulate kernel, 7 in the fold, 7 in the planes) against ~90 lane-wise
// ones of 7.3 us + the accumulation before.
// A list longer than its ch slots (several WINDOWS of one scalar carrying the same digit, in every term of a chunk: not a witness anybody
// has) raises a flag; msm_finish_batch then runs the ordinary pipeline on the same inputs — never a wrong result.
// No MFMA (256-bit modular integers); bound by the latency of the EC addition chains.
#include "msm_shape.h"
#include "msm.h"
#include "ec29_quad_dev.h"

namespace plk {

constexpr uint32_t SM_BUCKETS = 512;             // index v in [1, 255] = lo value v, index 255 + v = hi value v in [1, 256]
constexpr uint32_t SM_THREADS = 256;             // a workgroup owns 64 buckets of its terms, FOUR lanes per bucket; one or two waves per SIMD
// Terms per workgroup (= slots of a bucket's list: a constant column fills exactly that many): 64, 128 or 256, chosen per launch so that the grid is about one
// or two workgroups per CU.  The kernel is bound by the additions a SIMD issues (~5.6 us per wave and addition; a lone wave issues at the SIMD's rate), so what
// counts is the LONGEST list share of a wave.  Third layout of the round: (1) one lane per bucket: the longest of 65 536 lists of a 2^12-term commitment held
// 10-11 entries against a mean of 1.9 and WAS the kernel's duration (73-80 us); (2) two lanes per bucket, an entry joins the shorter list: 50 us, a batch of four
// 107-145 us; (3) this one — ONE list per bucket, split evenly over four lanes (lane j takes entries j, j + 4, ..): inside a bucket the balance is exact, and
// with 4x the terms per bucket the lists are longer and relatively more even.
constexpr uint32_t SM_CH_MIN = 64, SM_CH_MAX = 256, SM_WG_BUCKETS = 64, SM_LANES_PER_BUCKET = 4;
static size_t sm_lds(uint32_t ch) { return (size_t)(SM_WG_BUCKETS + SM_WG_BUCKETS * (ch + 1)) * sizeof(uint32_t); }

__device__ __forceinline__ void small_chain_priority() { __builtin_amdgcn_s_setprio(3); }

// grid (8 G, batch): workgroup 8 g + s takes terms [g ch, (g + 1) ch) and an EIGHTH of the bucket space — s & 1: lo / hi values, s >> 1: which 64 of the 256 values.
// bases = copy 0 of the fixed-base table at the commitment's first point; copy w lies w * copy_stride points on.
// QUADSUM (the default; PLK_MSM_SMALL_QUADSUM=0 is the A/B knob): the four lanes of a bucket add their sums up before they store (three four-lane additions
// on the end of the kernel): a quarter of the partial sums for msm_small_fold, whose quads walk G2 / Q of them one after the other.
template <bool QUADSUM>
__global__ void __launch_bounds__(SM_THREADS, 2) msm_small_accumulate(const G1Affine *bases, ScalarSet set, uint32_t n, uint32_t ch, uint32_t copy_stride,
                                                                     XyzzW *partials, uint32_t *flag) {
    extern __shared__ uint32_t sm_lds_mem[];
    uint32_t *cnt = sm_lds_mem, *list = sm_lds_mem + SM_WG_BUCKETS;
    const uint32_t tid = threadIdx.x, g = blockIdx.x >> 3, hi_set = blockIdx.x & 1, eighth = (blockIdx.x >> 1) & 3, m = blockIdx.y, first = g * ch;
    const uint32_t stride = ch + 1;                            // (odd stride: the lanes' reads fall into different banks)
    if (tid < SM_WG_BUCKETS) cnt[tid] = 0;
    __syncthreads();
    if (tid < ch && first + tid < n) {
        int32_t d[RC_WINDOWS];
        recode17(to_canonical(load_fp(set.v[m] + first + tid)), d);
#pragma unroll
        for (uint32_t w = 0; w < RC_WINDOWS; w++) {
            if (!d[w]) continue;
            const uint32_t mg = (uint32_t)(d[w] < 0 ? -d[w] : d[w]);
            const uint32_t v = hi_set ? mg >> 8 : mg & 255u;                  // lo in [0, 255], hi in [0, 256]; 0 = no entry in this set
            if (!v) continue;
            const uint32_t b = hi_set ? v - 1 : v;                           // bucket index inside the set
            if ((b >> 6) != eighth) continue;
            const uint32_t t = b & 63u, pos = atomicAdd(&cnt[t], 1u);
            if (pos < ch) list[t * stride + pos] = (d[w] < 0 ? 0x80000000u : 0u) | (w << 8) | tid;
        }
    }
    __syncthreads();
    const uint32_t bucket = tid >> 2, sub = tid & 3;
    uint32_t c = cnt[bucket];
    if (c > ch) { if (sub == 0) atomicOr(flag, 1u); c = ch; }
    const uint32_t *mine = list + bucket * stride;
    auto point = [&](uint32_t e) __attribute__((always_inline)) { return bases + (size_t)((e >> 8) & 15u) * copy_stride + first + (e & 255u); };
    XyzzW acc = xyzzw_identity();
    uint32_t e_next = sub < c ? mine[sub] : 0;
    G1Affine nx;
    if (sub < c) nx = load_affine(point(e_next));
    for (uint32_t r = sub; r < c; r += SM_LANES_PER_BUCKET) {  // (the next point is requested before the addition that hides its latency)
        const G1Affine cur = nx;
        const bool neg = (e_next >> 31) != 0;
        if (r + SM_LANES_PER_BUCKET < c) { e_next = mine[r + SM_LANES_PER_BUCKET]; nx = load_affine(point(e_next)); }
        AffW q; q.x = unpack<FqW>(cur.x); q.y = unpack<FqW>(cur.y);
        xyzzw_add_mixed(acc, q, neg);
    }
    if (QUADSUM) {
        // partial sums: [m][g][512 buckets]; the four sums of the quad in distributed form (ec29_quad_dev.h), three additions, each lane stores its coordinate
        const FqW9 c1 = quad_distribute<1>(acc, sub), c2 = quad_distribute<2>(acc, sub), c3 = quad_distribute<3>(acc, sub);
        FqW9 X = quad_distribute<0>(acc, sub);
        for (int k = 1; k < 4; k++) X = xyzzw_add_dist(X, wsel(k == 1, c1, wsel(k == 2, c2, c3)), sub);     // (one addition site)
        store_coord(partials + ((size_t)m * (gridDim.x >> 3) + g) * SM_BUCKETS + hi_set * 256 + eighth * 64 + bucket, sub, X);
    } else {
        // partial sums: [m][4 g + sub][512 buckets] — msm_small_fold sees 4 G "workgroups"
        store_xyzzw(partials + ((size_t)m * (gridDim.x >> 3) * 4 + 4 * g + sub) * SM_BUCKETS + hi_set * 256 + eighth * 64 + bucket, acc);
    }
}

// The two tree kernels run their full additions four lanes at a time (ec29_quad_dev.h: a quad of lanes shares one addition, four products
// deep instead of fourteen), since late round 6 in the DISTRIBUTED form: lane r of a quad holds only coordinate r of the running sum.
// One addition site per kernel (the operand is chosen beforehand).  Both are sized for ONE wave per SIMD (<= 1024 waves on the chip): a lone wave
// already issues at its SIMD's rate (tools/ubench_lanes), a second wave on the SIMD doubles the time of every tree level — measured 4.5 us per level
// with two waves per SIMD (planes of 512 threads, folds of 2048 waves) against 2.3 us alone.
//
// buckets[m][b] = sum of the G2 partial sums of bucket b.  Q = 2^QL quads per bucket, chosen by the launch so that batch * 512 * Q quads are about 1024
// waves: 32 quads for one commitment (two waves per bucket, the last level through LDS), 16 / 8 / 4 for batches of 2 / 3-4 / 5-8.  Quad q of a bucket takes
// partial sums q, q + Q, .. one after the other, then a tree over the bucket's quads.  Workgroup = 128 threads = 32 quads = 32 / Q buckets.
template <uint32_t QL>
__global__ void __launch_bounds__(128) msm_small_fold(const XyzzW *partials, uint32_t G2, XyzzW *buckets) {
    constexpr uint32_t Q = 1u << QL, BPB = 32u >> QL;
    __shared__ __attribute__((aligned(16))) uint32_t sh[4][9];
    small_chain_priority();
    const uint32_t tid = threadIdx.x, lane = tid & 63, role = tid & 3, m = blockIdx.y, quad = (tid >> 2) & (Q - 1);
    const uint32_t b = blockIdx.x * BPB + (tid >> (2 + QL));
    const XyzzW *P = partials + (size_t)m * G2 * SM_BUCKETS + b;
    const uint32_t nseq = (G2 + Q - 1) / Q, total = nseq + QL;
    FqW9 X = w_zero<FqW>();                                    // (distributed form: lane `role` of the quad holds coordinate `role`; all zero = the identity)
    for (uint32_t step = 0; step < total; step++) {
        FqW9 O = w_zero<FqW>();
        if (step < nseq) {
            const uint32_t g = quad + Q * step;
            if (g < G2) O = load_coord(P + (size_t)g * SM_BUCKETS, role);
        } else {
            const uint32_t k = step - nseq;
            if (QL == 5 && k == 4) {                          // the two waves' sums change hands through LDS
                if (tid >= 64 && lane < 4) for (int i = 0; i < 9; i++) sh[role][i] = X.l[i];
                __syncthreads();
                if (tid < 4) for (int i = 0; i < 9; i++) O.l[i] = sh[role][i];
            } else O = coord_shfl_xor(X, 4 << k);
        }
        X = xyzzw_add_dist(X, O, role);                       // the one addition site of the kernel
    }
    if (quad == 0) store_coord(buckets + (size_t)m * SM_BUCKETS + b, role, X);
}

// grid (17, batch), 256 threads = 64 quads: one wave per SIMD of its CU.  Plane p < 8: bit p of the lo value; plane 8 + b: bit b of the hi value (b = 8: the
// one bucket hi = 256).  Quad q takes the q-th and the (q + 64)-th bucket of the plane; tree over the 16 quads of a wave, then over the four waves.
constexpr uint32_t SM_PLANES = 17;
__global__ void __launch_bounds__(256) msm_small_planes(const XyzzW *buckets, G1Xyzz *planes, const uint32_t *flag, uint32_t *flag_out, uint32_t extra) {
    __shared__ __attribute__((aligned(16))) uint32_t sh[4][4][9];
    small_chain_priority();
    const uint32_t p = blockIdx.x, m = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, role = tid & 3, quad = tid >> 2;
    const bool hi = p >= 8;
    const uint32_t bit = hi ? p - 8 : p;
    const XyzzW *B = buckets + (size_t)m * SM_BUCKETS;
    auto bucket_of = [&](uint32_t j) {                        // the j-th of the 128 values of [1, 255] with `bit` set
        const uint32_t v = ((j >> bit) << (bit + 1)) | (1u << bit) | (j & ((1u << bit) - 1));
        return B + (hi ? 255 + v : v);
    };
    FqW9 X = w_zero<FqW>();
    if (bit < 8) X = load_coord(bucket_of(quad), role);
    else if (quad == 0) X = load_coord(B + 511, role);
    for (uint32_t k = 0; k < 7 + extra; k++) {                 // (extra: measurement knob PLK_MSM_SMALL_EXTRA — more levels that add the identity)
        FqW9 O = w_zero<FqW>();
        if (k == 0) { if (bit < 8) O = load_coord(bucket_of(quad + 64), role); }
        else if (k < 5) O = coord_shfl_xor(X, 4 << (k - 1));
        else if (k < 7) {
            if (k == 5) {
                if (lane < 4) for (int i = 0; i < 9; i++) sh[wave][role][i] = X.l[i];
                __syncthreads();
                if ((lane >> 2) < 4) { for (int i = 0; i < 9; i++) X.l[i] = sh[lane >> 2][role][i]; } else X = w_zero<FqW>();
            }
            O = coord_shfl_xor(X, 4 << (k - 5));
        }
        X = xyzzw_add_dist(X, O, role);                       // the one addition site of the kernel
    }
    if (tid < 4) {                                            // each lane of the first quad exports its coordinate (external form: canonical, R = 2^256; the identity is all zero)
        const bool inf = quad_flag<2>(w_all_zero(X));
        Fq *out = &planes[(size_t)m * SM_PLANES + p].x + role;
        store_fp(out, inf ? Fq::zero() : pack<FqParams>(s_from_w(X)));
    }
    if (tid == 0 && p == 0 && m == 0) *flag_out = *flag;
}

// enqueues the three launches on `stream`.  The seventeen points per commitment and the overflow flag (one uint32 behind them) are stored by the
// last kernel straight into `host_out` (page-locked, device-visible): no copy launch on the tail of a chain this short.
int32_t msm_small_launch(plk_ctx::MsmSlot &S, hipStream_t stream, const G1Affine *bases, uint32_t copy_stride, const ScalarSet &set,
                         uint32_t batch, uint32_t n, bool ev_on, void *host_out) {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_small_accumulate<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_lds(SM_CH_MAX)));
        PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_small_accumulate<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_lds(SM_CH_MAX)));
        attr_set = true;
    }
    static const uint32_t probe_ch = [] { const char *e = getenv("PLK_MSM_SMALL_CH"); return e ? (uint32_t)atoi(e) : 0u; }();   // A/B knob: 64, 128 or 256
    static const uint32_t probe_extra = [] { const char *e = getenv("PLK_MSM_SMALL_EXTRA"); return e ? (uint32_t)atoi(e) : 0u; }();
    static const uint32_t want_wgs = [] { const char *e = getenv("PLK_MSM_SMALL_WGS"); return e ? (uint32_t)atoi(e) : 512u; }();
    uint32_t ch = SM_CH_MIN;
    while (ch < SM_CH_MAX && (uint64_t)batch * 8 * ((n + ch - 1) / ch) > want_wgs) ch <<= 1;
    if (probe_ch == 64 || probe_ch == 128 || probe_ch == 256) ch = probe_ch;
    const uint32_t G = (n + ch - 1) / ch;
    static const int probe_quadsum = [] { const char *e = getenv("PLK_MSM_SMALL_QUADSUM"); return e ? atoi(e) : -1; }();      // A/B knob: 0 never, 1 always
    const bool quadsum = probe_quadsum != 0;
    const uint32_t G2 = quadsum ? G : 4 * G;
    PLK_TRY(S.e.reserve((size_t)batch * 4 * G * SM_BUCKETS * sizeof(XyzzW)));
    PLK_TRY(S.c.reserve((size_t)batch * SM_BUCKETS * sizeof(XyzzW) + 16));
    XyzzW *partials = S.e.as<XyzzW>(), *buckets = S.c.as<XyzzW>();
    uint32_t *flag = reinterpret_cast<uint32_t *>(buckets + (size_t)batch * SM_BUCKETS);
    G1Xyzz *planes = static_cast<G1Xyzz *>(host_out);
    PLK_HIP(hipMemsetAsync(flag, 0, 16, stream));
    if (ev_on) PLK_HIP(hipEventRecord(S.ev[0], stream));
    if (quadsum) hipLaunchKernelGGL(msm_small_accumulate<true>, dim3(8 * G, batch), dim3(SM_THREADS), sm_lds(ch), stream, bases, set, n, ch, copy_stride, partials, flag);
    else hipLaunchKernelGGL(msm_small_accumulate<false>, dim3(8 * G, batch), dim3(SM_THREADS), sm_lds(ch), stream, bases, set, n, ch, copy_stride, partials, flag);
    if (ev_on) (void)hipEventRecord(S.ev[1], stream);
    (void)hipEventRecord(S.acc_done, stream);
    static const int probe_ql = [] { const char *e = getenv("PLK_MSM_SMALL_FOLD_QL"); return e ? atoi(e) : 0; }();               // A/B knob: 2 .. 5
    const uint32_t ql = (probe_ql >= 2 && probe_ql <= 5) ? (uint32_t)probe_ql : (batch == 1 ? 5u : batch == 2 ? 4u : batch <= 4 ? 3u : 2u);
    const dim3 fgrid(SM_BUCKETS / (32u >> ql), batch);
    if (ql == 5) hipLaunchKernelGGL(msm_small_fold<5>, fgrid, dim3(128), 0, stream, (const XyzzW *)partials, G2, buckets);
    else if (ql == 4) hipLaunchKernelGGL(msm_small_fold<4>, fgrid, dim3(128), 0, stream, (const XyzzW *)partials, G2, buckets);
    else if (ql == 3) hipLaunchKernelGGL(msm_small_fold<3>, fgrid, dim3(128), 0, stream, (const XyzzW *)partials, G2, buckets);
    else hipLaunchKernelGGL(msm_small_fold<2>, fgrid, dim3(128), 0, stream, (const XyzzW *)partials, G2, buckets);
    hipLaunchKernelGGL(msm_small_planes, dim3(SM_PLANES, batch), dim3(256), 0, stream, (const XyzzW *)buckets, planes, (const uint32_t *)flag,
                       reinterpret_cast<uint32_t *>(planes + (size_t)batch * SM_PLANES), probe_extra);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

}  // namespace plk
