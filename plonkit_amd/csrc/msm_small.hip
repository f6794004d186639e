// Short commitments (round 6): a KZG commitment of <= 2^15 terms in three short launches.
//
// The pipeline of msm.hip is tuned for 2^20 terms: 2^16 buckets, eight launches, and a bucket reduction that is a chain of ~90
// dependent full additions — at 2^12..2^14 terms a commitment costs 0.46-0.63 ms of which 0.35 ms is that chain, and the
// reference's own CI circuit (test/test_poseidon_plonk.sh: a 2^12 domain) spends 80 % of a proof there.  Same group element as
// commit_using_monomials -> dense_multiexp (src/plonk.rs:152-159), other schedule:
//   * the fixed-base table of msm.hip stays (copy w holds 2^(17w) * P_i), so all 15 signed 17-bit windows of a commitment share
//     one bucket space; a digit's magnitude m <= 2^16 is split m = 256 * hi + lo and the entry is dropped into TWO small bucket
//     sets — lo in [1, 255], hi in [1, 256] — : 30 mixed additions per term instead of 15, into 511 buckets instead of 65536;
//   * msm_small_accumulate: a workgroup of 512 lanes takes 64 terms; lane b OWNS bucket b.  One wave recodes the 64 scalars and
//     appends every entry to the lists of its two buckets (LDS atomics), then every lane adds up its own list: ~4 mixed additions
//     per lane for uniform scalars (a constant column — 64 equal scalars — makes one lane add 64: still shorter than the old tail);
//   * msm_small_fold: the G = n / 64 workgroups' sums of a bucket, one wave per bucket (shuffle tree of full additions);
//   * msm_small_planes: sum_j j * B_j = sum_b 2^b * (sum of the B_j with bit b of j set): seventeen plain tree sums over <= 128
//     buckets, side by side; the seventeen points go to the host, whose Horner (16 doublings + 16 additions) takes ~15 us.
// The longest dependent chain is ~10 mixed + 13 full additions (against ~90 + the accumulation before).
// A list longer than its 64 slots (several WINDOWS of one scalar carrying the same digit, 64 times over: not a witness anybody
// has) raises a flag; msm_finish_batch then runs the ordinary pipeline on the same inputs — never a wrong result.
// No MFMA (256-bit modular integers); bound by the latency of the EC addition chains.
#include "msm_shape.h"
#include "msm.h"

namespace plk {

constexpr uint32_t SM_THREADS = 512;             // = buckets of a workgroup: index v in [1, 255] = lo value v, index 255 + v = hi value v in [1, 256]
constexpr uint32_t SM_CH = 64;                   // terms per workgroup
constexpr uint32_t SM_CAP = 64, SM_STRIDE = SM_CAP + 1;      // list slots per bucket (odd stride: the lanes' reads fall into different banks)
constexpr size_t SM_LDS = (size_t)(SM_THREADS + SM_THREADS * SM_STRIDE) * sizeof(uint32_t);

__device__ __forceinline__ void small_chain_priority() { __builtin_amdgcn_s_setprio(3); }

__device__ __forceinline__ XyzzW sm_shfl_xor(const XyzzW &v, int mask) {
    XyzzW r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.x.l[i] = __shfl_xor(v.x.l[i], mask);
        r.y.l[i] = __shfl_xor(v.y.l[i], mask);
        r.zz.l[i] = __shfl_xor(v.zz.l[i], mask);
        r.zzz.l[i] = __shfl_xor(v.zzz.l[i], mask);
    }
    return r;
}

// grid (G, batch).  bases = copy 0 of the fixed-base table at the commitment's first point; copy w lies w * copy_stride points on.
__global__ void __launch_bounds__(SM_THREADS, 1) msm_small_accumulate(const G1Affine *bases, ScalarSet set, uint32_t n, uint32_t copy_stride,
                                                                     XyzzW *partials, uint32_t *flag) {
    extern __shared__ uint32_t sm_lds[];
    uint32_t *cnt = sm_lds, *list = sm_lds + SM_THREADS;
    const uint32_t tid = threadIdx.x, g = blockIdx.x, m = blockIdx.y, first = g * SM_CH;
    cnt[tid] = 0;
    __syncthreads();
    if (tid < SM_CH && first + tid < n) {
        int32_t d[RC_WINDOWS];
        recode17(to_canonical(load_fp(set.v[m] + first + tid)), d);
#pragma unroll
        for (uint32_t w = 0; w < RC_WINDOWS; w++) {
            if (!d[w]) continue;
            const uint32_t mg = (uint32_t)(d[w] < 0 ? -d[w] : d[w]), lo = mg & 255u, hi = mg >> 8;
            const uint32_t e = (d[w] < 0 ? 0x80000000u : 0u) | (w << 8) | tid;
            if (lo) { const uint32_t pos = atomicAdd(&cnt[lo], 1u); if (pos < SM_CAP) list[lo * SM_STRIDE + pos] = e; }
            if (hi) { const uint32_t pos = atomicAdd(&cnt[255 + hi], 1u); if (pos < SM_CAP) list[(255 + hi) * SM_STRIDE + pos] = e; }
        }
    }
    __syncthreads();
    uint32_t c = cnt[tid];
    if (c > SM_CAP) { atomicOr(flag, 1u); c = SM_CAP; }
    const uint32_t *mine = list + tid * SM_STRIDE;
    auto point = [&](uint32_t e) __attribute__((always_inline)) { return bases + (size_t)((e >> 8) & 15u) * copy_stride + first + (e & 63u); };
    XyzzW acc = xyzzw_identity();
    uint32_t e_next = c ? mine[0] : 0;
    G1Affine nx;
    if (c) nx = load_affine(point(e_next));
    for (uint32_t r = 0; r < c; r++) {                        // (the next point is requested before the addition that hides its latency)
        const G1Affine cur = nx;
        const bool neg = (e_next >> 31) != 0;
        if (r + 1 < c) { e_next = mine[r + 1]; nx = load_affine(point(e_next)); }
        AffW q; q.x = unpack<FqW>(cur.x); q.y = unpack<FqW>(cur.y);
        xyzzw_add_mixed(acc, q, neg);
    }
    store_xyzzw(partials + ((size_t)m * gridDim.x + g) * SM_THREADS + tid, acc);
}

// grid (128, batch), 256 threads: wave -> bucket b = 4 * blockIdx.x + wave; buckets[m][b] = sum over the G workgroups.
// One addition site (operands chosen beforehand), as in msm_task_reduce: two 144-byte points through an out-of-line call cost more than the arithmetic.
__global__ void __launch_bounds__(256) msm_small_fold(const XyzzW *partials, uint32_t G, XyzzW *buckets) {
    small_chain_priority();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x * 4 + wave, m = blockIdx.y;
    const XyzzW *P = partials + (size_t)m * G * SM_THREADS + b;
    XyzzW X = xyzzw_identity();
    uint32_t g = lane, step = 0;
    for (;;) {
        const bool loading = __any(g < G);                    // wave-uniform
        if (!loading && step == 6) break;
        XyzzW O = xyzzw_identity();
        if (loading) { if (g < G) { O = load_xyzzw(P + (size_t)g * SM_THREADS); g += 64; } }
        else { O = sm_shfl_xor(X, 1 << step); step++; }
        xyzzw_add(X, O);
    }
    if (lane == 0) store_xyzzw(buckets + (size_t)m * SM_THREADS + b, X);
}

// grid (17, batch), 128 threads.  Plane p < 8: bit p of the lo value; plane 8 + b: bit b of the hi value (b = 8: the one bucket hi = 256).
constexpr uint32_t SM_PLANES = 17;
__global__ void __launch_bounds__(128) msm_small_planes(const XyzzW *buckets, G1Xyzz *planes) {
    __shared__ __attribute__((aligned(16))) XyzzW sh;
    small_chain_priority();
    const uint32_t p = blockIdx.x, m = blockIdx.y, tid = threadIdx.x;
    const bool hi = p >= 8;
    const uint32_t bit = hi ? p - 8 : p;
    const XyzzW *B = buckets + (size_t)m * SM_THREADS;
    XyzzW X = xyzzw_identity();
    if (bit < 8) {
        const uint32_t v = ((tid >> bit) << (bit + 1)) | (1u << bit) | (tid & ((1u << bit) - 1));     // the 128 values of [1, 255] with `bit` set
        X = load_xyzzw(B + (hi ? 255 + v : v));
    } else if (tid == 0) X = load_xyzzw(B + 511);
    for (uint32_t step = 0; step < 7; step++) {
        XyzzW O;
        if (step < 6) O = sm_shfl_xor(X, 1 << step);
        else {
            if (tid == 64) sh = X;
            __syncthreads();
            O = tid < 64 ? sh : xyzzw_identity();
        }
        xyzzw_add(X, O);                                      // the one addition site of the kernel
    }
    if (tid == 0) store_xyzz(planes + (size_t)m * SM_PLANES + p, xyzzw_export(X));
}

// enqueues the three launches on `stream`; planes_out: batch * 17 points followed by the overflow flag (one uint32)
int32_t msm_small_launch(plk_ctx::MsmSlot &S, hipStream_t stream, const G1Affine *bases, uint32_t copy_stride, const ScalarSet &set,
                         uint32_t batch, uint32_t n, bool ev_on) {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_small_accumulate), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SM_LDS));
        attr_set = true;
    }
    const uint32_t G = (n + SM_CH - 1) / SM_CH;
    PLK_TRY(S.e.reserve((size_t)batch * G * SM_THREADS * sizeof(XyzzW)));
    PLK_TRY(S.c.reserve((size_t)batch * SM_THREADS * sizeof(XyzzW)));
    PLK_TRY(S.d.reserve((size_t)batch * SM_PLANES * sizeof(G1Xyzz) + 16));
    XyzzW *partials = S.e.as<XyzzW>(), *buckets = S.c.as<XyzzW>();
    G1Xyzz *planes = S.d.as<G1Xyzz>();
    uint32_t *flag = reinterpret_cast<uint32_t *>(planes + (size_t)batch * SM_PLANES);
    PLK_HIP(hipMemsetAsync(flag, 0, 16, stream));
    if (ev_on) PLK_HIP(hipEventRecord(S.ev[0], stream));
    hipLaunchKernelGGL(msm_small_accumulate, dim3(G, batch), dim3(SM_THREADS), SM_LDS, stream, bases, set, n, copy_stride, partials, flag);
    if (ev_on) (void)hipEventRecord(S.ev[1], stream);
    (void)hipEventRecord(S.acc_done, stream);
    hipLaunchKernelGGL(msm_small_fold, dim3(SM_THREADS / 4, batch), dim3(256), 0, stream, (const XyzzW *)partials, G, buckets);
    hipLaunchKernelGGL(msm_small_planes, dim3(SM_PLANES, batch), dim3(128), 0, stream, (const XyzzW *)buckets, planes);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

}  // namespace plk
