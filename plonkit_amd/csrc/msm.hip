// Pippenger multi-scalar multiplication into BN254 G1 for gfx950.
//
// Replaces bellman_ce::multiexp::dense_multiexp as reached through
// kate_commitment::commit_using_monomials — the 11 commitments of prove_by_steps
// (src/plonk.rs:152-159) and the 11 of make_verification_key (src/plonk.rs:122-124).
// The result is the same group element; the schedule is MI355X-first, not bellman's:
//   * signed c-bit windows (c <= 16): 2^(c-1) buckets per window, W = floor(254/c)+1 windows.
//   * scalars are taken out of Montgomery form and recoded on the fly (never stored).
//   * two-level bucket sort without a global sort: (1) a coarse partition by (window, top bits of
//     the bucket index) built with LDS histograms + one global reservation per (block, bin);
//     (2) each workgroup then owns one coarse bin = 128 consecutive buckets, counting-sorts its
//     entries inside LDS and accumulates them — two lanes per bucket keep XYZZ accumulators in
//     registers for the whole bin, SRS points are gathered as 64-byte coalesced affine records
//     straight from the HBM-resident SRS (it stays in the 256 MiB Infinity Cache at 2^20).
//   * a bucket that is hot inside a chunk (repeated scalars: all-ones, all -1) is reduced by the
//     whole workgroup with wave64 shuffles (__shfl_xor tree) instead of serially.
//   * per-window sum_b (b+1)*B_b by 16-bucket running sums + small scalar multiple, then a
//     workgroup tree; the last 255 doublings (Horner over windows) run on the host, where one
//     serial EC chain is 20x faster than on a GPU lane.
// No MFMA (256-bit modular integers), arithmetic-bound on v_mad_u64_u32; HBM traffic is the
// algorithmic 96 B/term plus 8 B/term/window of index lists.
#include "ctx.h"
#include "ec.cuh"
#include "msm.h"
#include "hostmath.h"
#include <cstring>

namespace plk {

constexpr int MSM_THREADS = 256;
constexpr uint32_t FINE_BITS = 7;                 // 128 buckets per accumulate workgroup
constexpr uint32_t FINE = 1u << FINE_BITS;
constexpr uint32_t CHUNK = 8192;                  // entries per accumulate workgroup (sorted in LDS)
constexpr uint32_t SCALARS_PER_BLOCK = 4096;      // partition kernels
constexpr uint32_t TASK_MAX = CHUNK;
constexpr uint32_t HEAVY = 320;                   // per-chunk bucket population handled cooperatively

struct MsmParams {
    uint32_t n;
    uint32_t c;               // window bits
    uint32_t windows;         // W
    uint32_t coarse_bits;     // c - 1 - FINE_BITS
    uint32_t nbins;           // 1 << coarse_bits
};

// -------------------------------------------------------------------------- scalar recoding
struct Digits {
    uint32_t limbs[8];        // canonical scalar
    uint32_t carry;
};

__device__ __forceinline__ uint32_t extract_bits(const uint32_t *k, uint32_t pos, uint32_t c) {
    uint32_t limb = pos >> 5, off = pos & 31;
    if (limb >= 8) return 0;
    uint64_t v = k[limb];
    if (limb + 1 < 8) v |= (uint64_t)k[limb + 1] << 32;
    return (uint32_t)(v >> off) & ((1u << c) - 1);
}

// next signed digit in (-2^(c-1), 2^(c-1)]; returns magnitude, sets neg
__device__ __forceinline__ uint32_t next_digit(Digits &d, uint32_t w, uint32_t c, bool &neg_out) {
    uint32_t v = extract_bits(d.limbs, w * c, c) + d.carry;
    if (v > (1u << (c - 1))) { d.carry = 1; neg_out = true; return (1u << c) - v; }
    d.carry = 0; neg_out = false; return v;
}

__device__ __forceinline__ Digits load_scalar(const Fr *scalars, uint32_t i) {
    Fr k = to_canonical(load_fp(scalars + i));
    Digits d;
#pragma unroll
    for (int j = 0; j < 8; j++) d.limbs[j] = k.l[j];
    d.carry = 0;
    return d;
}

// ------------------------------------------------------------------- coarse partition kernels
template <bool SCATTER>
__global__ void __launch_bounds__(MSM_THREADS) msm_partition(const Fr *scalars, MsmParams p, uint32_t *hist_or_cursor,
                                                              const uint32_t *bin_start, uint32_t *entries) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *lh = reinterpret_cast<uint32_t *>(smem);                 // [W * nbins] counts
    uint32_t *lbase = lh + p.windows * p.nbins;                        // [W * nbins] reserved bases (SCATTER)
    const uint32_t total_bins = p.windows * p.nbins;
    const uint32_t tid = threadIdx.x;
    const uint32_t first = blockIdx.x * SCALARS_PER_BLOCK;
    for (uint32_t b = tid; b < total_bins; b += MSM_THREADS) lh[b] = 0;
    __syncthreads();
    for (uint32_t i = first + tid; i < first + SCALARS_PER_BLOCK && i < p.n; i += MSM_THREADS) {
        Digits d = load_scalar(scalars, i);
        for (uint32_t w = 0; w < p.windows; w++) {
            bool ng; uint32_t m = next_digit(d, w, p.c, ng);
            if (m) atomicAdd(&lh[w * p.nbins + ((m - 1) >> FINE_BITS)], 1u);
        }
    }
    __syncthreads();
    if (!SCATTER) {
        for (uint32_t b = tid; b < total_bins; b += MSM_THREADS)
            if (lh[b]) atomicAdd(&hist_or_cursor[b], lh[b]);
        return;
    }
    for (uint32_t b = tid; b < total_bins; b += MSM_THREADS) {
        uint32_t cnt = lh[b];
        lbase[b] = cnt ? bin_start[b] + atomicAdd(&hist_or_cursor[b], cnt) : 0;
        lh[b] = 0;
    }
    __syncthreads();
    for (uint32_t i = first + tid; i < first + SCALARS_PER_BLOCK && i < p.n; i += MSM_THREADS) {
        Digits d = load_scalar(scalars, i);
        for (uint32_t w = 0; w < p.windows; w++) {
            bool ng; uint32_t m = next_digit(d, w, p.c, ng);
            if (m) {
                uint32_t bin = w * p.nbins + ((m - 1) >> FINE_BITS);
                uint32_t rank = atomicAdd(&lh[bin], 1u);
                entries[lbase[bin] + rank] = (i << 8) | (ng ? 0x80u : 0u) | ((m - 1) & (FINE - 1));
            }
        }
    }
}

// exclusive scan of the (W * nbins) histogram -> bin_start[total+1], and of the per-bin task counts
// ceil(count / TASK_MAX) -> task_start[total+1]; clears the cursors.  A bin that is much larger
// than the others (top window with few bits, repeated scalars) is cut into several tasks so that
// no single workgroup walks it alone.
__global__ void __launch_bounds__(1024) msm_scan_bins(uint32_t *hist, uint32_t *bin_start, uint32_t *task_start, uint32_t total_bins) {
    __shared__ uint32_t sums[1024], tsums[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (total_bins + 1023) / 1024;
    uint32_t lo = tid * per, hi = lo + per < total_bins ? lo + per : total_bins, s = 0, ts = 0;
    if (lo > total_bins) lo = total_bins;
    for (uint32_t i = lo; i < hi; i++) { s += hist[i]; ts += (hist[i] + TASK_MAX - 1) / TASK_MAX; }
    sums[tid] = s; tsums[tid] = ts;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = tid >= off ? sums[tid - off] : 0, tv = tid >= off ? tsums[tid - off] : 0;
        __syncthreads();
        sums[tid] += v; tsums[tid] += tv;
        __syncthreads();
    }
    uint32_t run = tid ? sums[tid - 1] : 0, trun = tid ? tsums[tid - 1] : 0;
    for (uint32_t i = lo; i < hi; i++) {
        uint32_t c = hist[i];
        bin_start[i] = run; run += c;
        task_start[i] = trun; trun += (c + TASK_MAX - 1) / TASK_MAX;
        hist[i] = 0;
    }
    if (tid == 1023) { bin_start[total_bins] = sums[1023]; task_start[total_bins] = tsums[1023]; }
}

// --------------------------------------------------------------------- bucket accumulation
__device__ __forceinline__ G1Xyzz shfl_xor_xyzz(const G1Xyzz &v, int mask) {
    G1Xyzz r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.x.l[i] = __shfl_xor(v.x.l[i], mask);
        r.y.l[i] = __shfl_xor(v.y.l[i], mask);
        r.zz.l[i] = __shfl_xor(v.zz.l[i], mask);
        r.zzz.l[i] = __shfl_xor(v.zzz.l[i], mask);
    }
    return r;
}
__device__ __forceinline__ G1Xyzz shfl_down_xyzz(const G1Xyzz &v, int delta) {
    G1Xyzz r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.x.l[i] = __shfl_down(v.x.l[i], delta);
        r.y.l[i] = __shfl_down(v.y.l[i], delta);
        r.zz.l[i] = __shfl_down(v.zz.l[i], delta);
        r.zzz.l[i] = __shfl_down(v.zzz.l[i], delta);
    }
    return r;
}

__device__ __noinline__ void xyzz_add_noinline(G1Xyzz &a, const G1Xyzz &b) { xyzz_add(a, b); }

__device__ __forceinline__ void accumulate_run(G1Xyzz &acc, const G1Affine *bases, const uint32_t *sorted, uint32_t lo, uint32_t hi) {
    if (lo >= hi) return;
    uint32_t e = sorted[lo];
    G1Affine pt = load_affine(bases + (e >> 8));
    for (uint32_t i = lo; i < hi; i++) {
        uint32_t e_cur = e;
        G1Affine cur = pt;
        if (i + 1 < hi) { e = sorted[i + 1]; pt = load_affine(bases + (e >> 8)); }   // prefetch the next gather
        xyzz_add_mixed(acc, cur, (e_cur & 0x80u) != 0);
    }
}

// One workgroup per task = one slice (<= CHUNK entries) of a (window, coarse bin): 128 buckets.
//  1. counting sort of the slice by fine bucket inside LDS
//  2. buckets ranked by population; lane pair p takes the p-th most populated bucket, half a run per
//     lane, so the lanes of a wave walk runs of (nearly) equal length and short waves retire early
//  3. buckets hotter than HEAVY are reduced by the whole workgroup with wave64 shuffle trees
//  4. epilogue on one wave: T = sum_f B_f and S = sum_f (f+1) B_f of the 128 bucket sums by a
//     shuffle suffix scan — the task leaves only these two points behind
__global__ void __launch_bounds__(MSM_THREADS) msm_accumulate(const G1Affine *bases, const uint32_t *entries,
                                                               const uint32_t *bin_start, const uint32_t *task_start,
                                                               G1Xyzz *task_out, uint32_t *task_bin, MsmParams p) {
    __shared__ uint32_t sorted[CHUNK];
    __shared__ uint32_t cnt[FINE], start[FINE + 1], cursor[FINE], order[FINE], rank_of[FINE];
    __shared__ __attribute__((aligned(16))) G1Xyzz B[FINE];
    __shared__ __attribute__((aligned(16))) G1Xyzz wave_part[MSM_THREADS / 64];
    __shared__ uint32_t heavy_list[FINE], heavy_n;
    const uint32_t tid = threadIdx.x, task = blockIdx.x;
    const uint32_t total_bins = p.windows * p.nbins;
    if (task >= task_start[total_bins]) return;
    uint32_t blo = 0, bhi = total_bins;                       // bin = last index with task_start[bin] <= task
    while (bhi - blo > 1) { uint32_t mid = (blo + bhi) >> 1; if (task_start[mid] <= task) blo = mid; else bhi = mid; }
    const uint32_t bin = blo, slice = task - task_start[bin];
    const uint32_t bs = bin_start[bin], be = bin_start[bin + 1];
    const uint32_t s = bs + slice * CHUNK, e = (s + CHUNK < be) ? s + CHUNK : be, nc = e - s;

    if (tid < FINE) { cnt[tid] = 0; cursor[tid] = 0; }
    if (tid == 0) heavy_n = 0;
    __syncthreads();
    for (uint32_t idx = tid; idx < nc; idx += MSM_THREADS) atomicAdd(&cnt[entries[s + idx] & (FINE - 1)], 1u);
    __syncthreads();
    if (tid < 64) {                                           // exclusive scan of 128 counts by one wave
        uint32_t a = cnt[2 * tid], b = cnt[2 * tid + 1], v = a + b;
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(v, off); if ((int)tid >= off) v += t; }
        uint32_t ex = v - (a + b);
        start[2 * tid] = ex; start[2 * tid + 1] = ex + a;
        if (tid == 63) start[FINE] = v;
    }
    if (tid >= 64 && tid < 64 + FINE) {                       // rank by population (descending, ties by index)
        uint32_t b = tid - 64, c = cnt[b], r = 0;
        for (uint32_t o = 0; o < FINE; o++) { uint32_t co = cnt[o]; r += (co > c) || (co == c && o < b); }
        order[r] = b; rank_of[b] = r;
        if (c > HEAVY) heavy_list[atomicAdd(&heavy_n, 1u)] = b;
    }
    __syncthreads();
    for (uint32_t idx = tid; idx < nc; idx += MSM_THREADS) {
        uint32_t en = entries[s + idx], f = en & (FINE - 1);
        sorted[start[f] + atomicAdd(&cursor[f], 1u)] = en;
    }
    __syncthreads();
    const uint32_t pair = tid >> 1, half = tid & 1, my_bucket = order[pair];
    G1Xyzz acc = xyzz_identity();
    {
        uint32_t b0 = start[my_bucket], n_b = cnt[my_bucket];
        if (n_b <= HEAVY) {
            uint32_t mid = b0 + (n_b + 1) / 2;
            accumulate_run(acc, bases, sorted, half ? mid : b0, half ? b0 + n_b : mid);
        }
    }
    const uint32_t hn = heavy_n;
    for (uint32_t hI = 0; hI < hn; hI++) {
        uint32_t hb = heavy_list[hI], b0 = start[hb], n_b = cnt[hb];
        uint32_t per = (n_b + MSM_THREADS - 1) / MSM_THREADS;
        uint32_t lo = b0 + tid * per, hi = lo + per;
        if (lo > b0 + n_b) lo = b0 + n_b;
        if (hi > b0 + n_b) hi = b0 + n_b;
        G1Xyzz part = xyzz_identity();
        accumulate_run(part, bases, sorted, lo, hi);
        for (int m = 1; m < 64; m <<= 1) { G1Xyzz o = shfl_xor_xyzz(part, m); xyzz_add_noinline(part, o); }
        if ((tid & 63) == 0) wave_part[tid >> 6] = part;
        __syncthreads();
        if (tid == 2 * rank_of[hb]) {
            for (int wv = 0; wv < MSM_THREADS / 64; wv++) { G1Xyzz o = wave_part[wv]; xyzz_add_noinline(acc, o); }
        }
        __syncthreads();
    }
    {
        G1Xyzz other = shfl_xor_xyzz(acc, 1);
        if (half == 0) { xyzz_add_noinline(acc, other); B[my_bucket] = acc; }
    }
    __syncthreads();
    if (tid < 64) {
        G1Xyzz r1 = B[2 * tid + 1], P = B[2 * tid];
        xyzz_add_noinline(P, r1);                             // pair total
        for (int off = 1; off < 64; off <<= 1) {              // inclusive suffix scan of the pair totals
            G1Xyzz o = shfl_down_xyzz(P, off);
            if ((int)tid + off < 64) xyzz_add_noinline(P, o);
        }
        G1Xyzz nxt = shfl_down_xyzz(P, 1);                    // suffix sum starting at bucket 2*tid + 2
        if (tid == 63) nxt = xyzz_identity();
        xyzz_add_noinline(r1, nxt);                           // suffix sum starting at bucket 2*tid + 1
        G1Xyzz V = P;
        xyzz_add_noinline(V, r1);
        for (int m = 1; m < 64; m <<= 1) { G1Xyzz o = shfl_xor_xyzz(V, m); xyzz_add_noinline(V, o); }
        if (tid == 0) {
            store_xyzz(task_out + 2 * (size_t)task, V);       // S = sum_f (f+1) B_f
            store_xyzz(task_out + 2 * (size_t)task + 1, P);   // T = sum_f B_f
            task_bin[task] = bin;
        }
    }
}

// ------------------------------------------------------------------------ window reduction
// one workgroup per window: W_w = sum_t S_t + 2^FINE_BITS * sum_c c * D_c,  D_c = sum of T_t over bin c
__global__ void __launch_bounds__(MSM_THREADS) msm_window_sums(const G1Xyzz *task_out, const uint32_t *task_start, G1Xyzz *window_out, uint32_t nbins) {
    __shared__ __attribute__((aligned(16))) G1Xyzz sh[MSM_THREADS];
    const uint32_t tid = threadIdx.x, w = blockIdx.x;
    G1Xyzz ssum = xyzz_identity(), d = xyzz_identity();
    if (tid < nbins) {
        uint32_t bin = w * nbins + tid;
        for (uint32_t t = task_start[bin]; t < task_start[bin + 1]; t++) {
            G1Xyzz sv = load_xyzz(task_out + 2 * (size_t)t), tv = load_xyzz(task_out + 2 * (size_t)t + 1);
            xyzz_add_noinline(ssum, sv);
            xyzz_add_noinline(d, tv);
        }
    }
    // inclusive suffix scan of D over the bins
    sh[tid] = d;
    __syncthreads();
    for (uint32_t off = 1; off < MSM_THREADS; off <<= 1) {
        G1Xyzz o = (tid + off < MSM_THREADS) ? sh[tid + off] : xyzz_identity();
        __syncthreads();
        if (tid + off < MSM_THREADS) { xyzz_add_noinline(d, o); sh[tid] = d; }
        __syncthreads();
    }
    // sum_c c*D_c = sum_{k>=1} suffix_k ; times 2^FINE_BITS ; plus the S terms
    G1Xyzz v = tid >= 1 ? d : xyzz_identity();
    for (uint32_t i = 0; i < FINE_BITS; i++) v = xyzz_double(v);
    xyzz_add_noinline(v, ssum);
    sh[tid] = v;
    __syncthreads();
    for (uint32_t off = MSM_THREADS / 2; off > 0; off >>= 1) {
        if (tid < off) { G1Xyzz o = sh[tid + off]; xyzz_add_noinline(v, o); sh[tid] = v; }
        __syncthreads();
    }
    if (tid == 0) store_xyzz(window_out + w, v);
}

// ------------------------------------------------------------------- tiny inputs: no buckets
__global__ void __launch_bounds__(MSM_THREADS) msm_naive(const G1Affine *bases, const Fr *scalars, uint32_t n, G1Xyzz *block_out) {
    __shared__ __attribute__((aligned(16))) G1Xyzz sh[MSM_THREADS];
    const uint32_t tid = threadIdx.x, i = blockIdx.x * MSM_THREADS + tid;
    G1Xyzz acc = xyzz_identity();
    if (i < n) {
        Fr k = to_canonical(load_fp(scalars + i));
        G1Affine pt = load_affine(bases + i);
        if (!k.is_zero() && !is_inf(pt)) {
            for (int bit = 253; bit >= 0; bit--) {
                acc = xyzz_double(acc);
                if ((k.l[bit >> 5] >> (bit & 31)) & 1) xyzz_add_mixed(acc, pt, false);
            }
        }
    }
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t off = MSM_THREADS / 2; off > 0; off >>= 1) {
        if (tid < off) { G1Xyzz o = sh[tid + off]; xyzz_add_noinline(acc, o); sh[tid] = acc; }
        __syncthreads();
    }
    if (tid == 0) store_xyzz(block_out + blockIdx.x, acc);
}

// ------------------------------------------------------------------------------ host side
static uint32_t pick_window_bits(uint64_t n) {
    if (n < (1u << 15)) return 12;
    if (n < (1u << 18)) return 14;
    return 16;
}

int32_t ensure_pinned(plk_ctx *ctx, size_t bytes);

int32_t msm_enqueue(plk_ctx *ctx, const Fr *scalars_dev, uint64_t n, uint64_t base_offset, hipStream_t stream) {
    if (!ctx->srs) { set_error("msm: no SRS uploaded (plk_srs_upload)"); return PLK_ERR_SRS; }
    if (base_offset + n > ctx->srs_n) { set_error("msm: SRS too small for this commitment"); return PLK_ERR_SRS; }
    if (n >= (1ull << 24) + 1) { set_error("msm: more than 2^24 terms per call (shard the commitment)"); return PLK_ERR_SIZE; }
    const G1Affine *bases = reinterpret_cast<const G1Affine *>(ctx->srs) + base_offset;
    ctx->msm_pending_parts = 0;
    ctx->msm_windows = 0;
    if (n == 0) return PLK_OK;
    if (n < 4096) {
        uint32_t blocks = (uint32_t)((n + MSM_THREADS - 1) / MSM_THREADS);
        PLK_TRY(ctx->msm_d.reserve(blocks * sizeof(G1Xyzz)));
        hipLaunchKernelGGL(msm_naive, dim3(blocks), dim3(MSM_THREADS), 0, stream, bases, scalars_dev, (uint32_t)n, ctx->msm_d.as<G1Xyzz>());
        PLK_HIP(hipGetLastError());
        ctx->msm_pending_parts = blocks;
        ctx->msm_c_bits = 0;
        PLK_TRY(ensure_pinned(ctx, blocks * sizeof(G1Xyzz)));
        PLK_HIP(hipMemcpyAsync(ctx->pinned, ctx->msm_d.p, blocks * sizeof(G1Xyzz), hipMemcpyDeviceToHost, stream));
        return PLK_OK;
    }
    MsmParams p;
    p.n = (uint32_t)n;
    p.c = pick_window_bits(n);
    p.windows = 254 / p.c + 1;
    p.coarse_bits = p.c - 1 - FINE_BITS;
    p.nbins = 1u << p.coarse_bits;
    const uint32_t total_bins = p.windows * p.nbins;
    const uint32_t max_tasks = total_bins + (uint32_t)(((uint64_t)p.windows * n) / TASK_MAX) + 1;
    PLK_TRY(ctx->msm_a.reserve((size_t)(3 * total_bins + 4 + max_tasks) * sizeof(uint32_t)));   // hist/cursor, bin_start, task_start, task_bin
    PLK_TRY(ctx->msm_b.reserve((size_t)p.windows * n * sizeof(uint32_t)));                       // entries
    PLK_TRY(ctx->msm_c.reserve((size_t)max_tasks * 2 * sizeof(G1Xyzz)));                         // per-task (S, T)
    PLK_TRY(ctx->msm_d.reserve((size_t)p.windows * sizeof(G1Xyzz)));                             // window sums
    uint32_t *hist = ctx->msm_a.as<uint32_t>(), *bin_start = hist + total_bins, *task_start = bin_start + total_bins + 1;
    uint32_t *task_bin = task_start + total_bins + 1;
    uint32_t *entries = ctx->msm_b.as<uint32_t>();
    G1Xyzz *task_out = ctx->msm_c.as<G1Xyzz>(), *window_out = ctx->msm_d.as<G1Xyzz>();

    PLK_HIP(hipMemsetAsync(hist, 0, total_bins * sizeof(uint32_t), stream));
    const uint32_t pblocks = (uint32_t)((n + SCALARS_PER_BLOCK - 1) / SCALARS_PER_BLOCK);
    const size_t plds = (size_t)2 * total_bins * sizeof(uint32_t);
    hipLaunchKernelGGL(msm_partition<false>, dim3(pblocks), dim3(MSM_THREADS), plds, stream, scalars_dev, p, hist, (const uint32_t *)nullptr, (uint32_t *)nullptr);
    hipLaunchKernelGGL(msm_scan_bins, dim3(1), dim3(1024), 0, stream, hist, bin_start, task_start, total_bins);
    hipLaunchKernelGGL(msm_partition<true>, dim3(pblocks), dim3(MSM_THREADS), plds, stream, scalars_dev, p, hist, (const uint32_t *)bin_start, entries);
    if (ctx->ev_on) PLK_HIP(hipEventRecord(ctx->ev[0], stream));
    hipLaunchKernelGGL(msm_accumulate, dim3(max_tasks), dim3(MSM_THREADS), 0, stream, bases, (const uint32_t *)entries, (const uint32_t *)bin_start,
                       (const uint32_t *)task_start, task_out, task_bin, p);
    if (ctx->ev_on) PLK_HIP(hipEventRecord(ctx->ev[1], stream));
    hipLaunchKernelGGL(msm_window_sums, dim3(p.windows), dim3(MSM_THREADS), 0, stream, (const G1Xyzz *)task_out, (const uint32_t *)task_start, window_out, p.nbins);
    PLK_HIP(hipGetLastError());
    PLK_TRY(ensure_pinned(ctx, p.windows * sizeof(G1Xyzz)));
    PLK_HIP(hipMemcpyAsync(ctx->pinned, window_out, p.windows * sizeof(G1Xyzz), hipMemcpyDeviceToHost, stream));
    ctx->msm_windows = p.windows;
    ctx->msm_c_bits = p.c;
    return PLK_OK;
}

static host::HJac xyzz_host_to_jac(const uint64_t *v) {
    using namespace host;
    HFq x, y, zz, zzz;
    memcpy(x.l, v, 32); memcpy(y.l, v + 4, 32); memcpy(zz.l, v + 8, 32); memcpy(zzz.l, v + 12, 32);
    if (zz.is_zero()) return HJac::inf();
    HJac r;
    HFq t2 = zzz.sqr(), zz2 = zz.sqr();
    r.x = x * zz * t2;
    r.y = y * zz2 * zz * t2;
    r.z = zz * zzz;
    return r;
}

// waits for the stream, then folds the window sums (Horner, c doublings per window) on the host
int32_t msm_finish(plk_ctx *ctx, hipStream_t stream, host::HJac *out) {
    using namespace host;
    PLK_HIP(hipStreamSynchronize(stream));
    const uint64_t *raw = reinterpret_cast<const uint64_t *>(ctx->pinned);
    HJac acc = HJac::inf();
    if (ctx->msm_windows) {
        for (int w = (int)ctx->msm_windows - 1; w >= 0; w--) {
            for (uint32_t i = 0; i < ctx->msm_c_bits; i++) acc = jac_double(acc);
            acc = jac_add(acc, xyzz_host_to_jac(raw + 16 * w));
        }
    } else {
        for (uint32_t i = 0; i < ctx->msm_pending_parts; i++) acc = jac_add(acc, xyzz_host_to_jac(raw + 16 * i));
    }
    *out = acc;
    return PLK_OK;
}

}  // namespace plk

using namespace plk;

extern "C" {

int32_t plk_msm_g1_enqueue_dev(plk_ctx *ctx, const void *scalars_dev, uint64_t n, uint64_t base_offset, void *stream) {
    if (!ctx || (!scalars_dev && n)) { set_error("plk_msm_g1: bad argument"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    ctx->msm_stream = stream ? (hipStream_t)stream : ctx->stream;
    return msm_enqueue(ctx, (const Fr *)scalars_dev, n, base_offset, ctx->msm_stream);
}

int32_t plk_msm_g1_finish(plk_ctx *ctx, plk_g1_jacobian *out) {
    if (!ctx || !out) { set_error("plk_msm_g1_finish: bad argument"); return PLK_ERR_ARG; }
    host::HJac j;
    PLK_TRY(msm_finish(ctx, ctx->msm_stream ? ctx->msm_stream : ctx->stream, &j));
    memcpy(out->x, j.x.l, 32); memcpy(out->y, j.y.l, 32); memcpy(out->z, j.z.l, 32);
    return PLK_OK;
}

int32_t plk_msm_g1_partial_dev(plk_ctx *ctx, const void *scalars_dev, uint64_t n, uint64_t base_offset, plk_g1_jacobian *out, void *stream) {
    PLK_TRY(plk_msm_g1_enqueue_dev(ctx, scalars_dev, n, base_offset, stream));
    return plk_msm_g1_finish(ctx, out);
}

int32_t plk_msm_g1_dev(plk_ctx *ctx, const void *scalars_dev, uint64_t n, uint64_t base_offset, plk_g1_affine *out, void *stream) {
    if (!out) { set_error("plk_msm_g1: null out"); return PLK_ERR_ARG; }
    plk_g1_jacobian j;
    PLK_TRY(plk_msm_g1_partial_dev(ctx, scalars_dev, n, base_offset, &j, stream));
    return plk_g1_sum_jacobian(&j, 1, out);
}

int32_t plk_msm_g1(plk_ctx *ctx, const plk_fr *scalars, uint64_t n, uint64_t base_offset, plk_g1_affine *out) {
    if (!ctx || (!scalars && n) || !out) { set_error("plk_msm_g1: bad argument"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(ctx->stage.reserve(n * sizeof(plk_fr) + 32));
    PLK_HIP(hipMemcpyAsync(ctx->stage.p, scalars, n * sizeof(plk_fr), hipMemcpyHostToDevice, ctx->stream));
    return plk_msm_g1_dev(ctx, ctx->stage.p, n, base_offset, out, nullptr);
}

int32_t plk_set_kernel_timing(plk_ctx *ctx, int32_t on) {
    if (!ctx) { set_error("null ctx"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    if (on && !ctx->ev[0]) { PLK_HIP(hipEventCreate(&ctx->ev[0])); PLK_HIP(hipEventCreate(&ctx->ev[1])); }
    ctx->ev_on = on != 0;
    return PLK_OK;
}

int32_t plk_msm_last_kernel_ms(plk_ctx *ctx, float *accumulate_ms) {
    if (!ctx || !accumulate_ms) { set_error("plk_msm_last_kernel_ms: bad argument"); return PLK_ERR_ARG; }
    if (!ctx->ev_on || !ctx->ev[0]) { set_error("kernel timing is off (plk_set_kernel_timing)"); return PLK_ERR_ARG; }
    PLK_HIP(hipEventSynchronize(ctx->ev[1]));
    PLK_HIP(hipEventElapsedTime(accumulate_ms, ctx->ev[0], ctx->ev[1]));
    return PLK_OK;
}

int32_t plk_g1_sum_jacobian(const plk_g1_jacobian *parts, uint64_t n, plk_g1_affine *out) {
    if ((!parts && n) || !out) { set_error("plk_g1_sum_jacobian: bad argument"); return PLK_ERR_ARG; }
    using namespace host;
    HJac acc = HJac::inf();
    for (uint64_t i = 0; i < n; i++) {
        HJac p; memcpy(p.x.l, parts[i].x, 32); memcpy(p.y.l, parts[i].y, 32); memcpy(p.z.l, parts[i].z, 32);
        acc = jac_add(acc, p);
    }
    HAffine a = jac_to_affine(acc);
    memcpy(out->x, a.x.l, 32); memcpy(out->y, a.y.l, 32);
    return PLK_OK;
}

}  // extern "C"
