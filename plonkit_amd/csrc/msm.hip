// Pippenger multi-scalar multiplication into BN254 G1 for gfx950.
//
// Replaces bellman_ce::multiexp::dense_multiexp as reached through
// kate_commitment::commit_using_monomials — the 11 commitments of prove_by_steps
// (src/plonk.rs:152-159) and the 11 of make_verification_key (src/plonk.rs:122-124).
// The result is the same group element; the schedule is MI355X-first, not bellman's:
//   * signed c-bit windows, W = floor(254/c)+1 of them, top window unsigned: c = 17 (15 windows, 2^16 buckets)
//     from 2^19 terms, 15 / 13 below; scalars leave Montgomery form once and are recoded into signed digits — in registers
//     when the commitment has ONE bucket set (msm_recode_count / msm_recode_scatter: the 2^20 shape), through an int32
//     digit array otherwise (msm_digits + msm_partition).
//   * the SRS is a FIXED base: a resident table holds 15 shifted copies 2^(17k) * P_i (0.94 GiB at 2^20 points,
//     built once per SRS).  Window w takes its point from copy w, so that all windows drop into ONE bucket set
//     (3 or 5 sets above 2^20 terms, where an entry can only address 5 or 3 copies) and the host Horner shrinks
//     to 17 * (sets - 1) doublings.
//   * two-level bucket sort without a global sort: (1) a coarse partition by the top bits of the bucket index:
//     per workgroup (1024 scalars x all windows, or a 16K-scalar chunk of one window) an LDS counting sort, one global
//     reservation per (block, bin) and contiguous copy-out of every bin's run; (2) a coarse bin (2^FB consecutive buckets; FB = 6: 1024 bins of ~15 K
//     entries at 2^20 terms, i.e. one task per bin) is cut into equal tasks of <= 16384 entries; a workgroup
//     counting-sorts its task inside LDS and cuts the sorted run into 256 EQUAL pieces, one per lane: one flat loop of
//     mixed additions with XYZZ accumulators in registers, a bucket boundary inside a piece only flushes the accumulator
//     (PRIMARY / HEAD / TAIL slots).  Points are gathered as 64-byte affine records.  All field arithmetic is the
//     carry-free 9x29-bit layer (field29_dev.h / ec29_dev.h): product-scanning products in lockstep pairs in the mixed
//     addition (throughput-bound), operand-scanning products in the full additions of the reduce kernels (latency-bound).
//   * a bucket spread over many lanes (repeated scalars: all-ones, all -1) is folded by a separate small kernel
//     (msm_fold_hot), a coarse bin spread over many tasks by another (msm_bin_fold).
//   * per task T = sum B_f and S = sum (f+1) B_f (running sums + 32-lane shuffle scan), then per bucket set
//     sum_t S_t, sum_t t * E_t and F_1..F_3 (msm_window_sums); the remaining shifts (2^FB, 2^8), sum u * F_u and the
//     Horner over the sets run on the host, where one serial EC chain is 20x faster than on a GPU lane.  The reduce
//     kernels are chains of full additions issued from one inlined call site each.
//   * up to 8 commitments over the same bases share every kernel launch (batch dimension), and three commitments
//     (or batches) may be in flight on three streams with their own scratch: the latency-bound reduction of one and
//     the digit / partition kernels of the one after next share the GPU with the accumulation in between.
// No MFMA (256-bit modular integers), bound by v_mad_u64_u32 issue; HBM sees the algorithmic 96 B/term plus the
// per-window gathers (64 B x W per term) from the table.
// Files: the accumulation kernel ("kernel A" above, step (2) of the sort) lives in msm_accumulate.hip — a translation unit
// of its own so that it can be compiled for ILP (plonkit_amd/build.py) —, the declarations both share in msm_shape.h;
// everything else (recoding, partition, bucket reduction, drivers, the FIFO of three slots, plk_ctx_share_srs) is here.
#include "msm_shape.h"
#include "msm.h"
#include "ec29_quad_dev.h"
#include "comm.h"
#include "hostmath.h"
#include <cstring>
#include <cstdlib>
#include <type_traits>

namespace plk {

// msm_small.hip: the short-commitment path
int32_t msm_small_launch(plk_ctx::MsmSlot &S, hipStream_t stream, const G1Affine *bases, uint32_t copy_stride, const ScalarSet &set,
                         uint32_t batch, uint32_t n, bool ev_on, void *host_out);
constexpr uint32_t SM_PLANES_HOST = 17;
// terms up to which a commitment takes it: 2^15 (measured, one at a time: 2^14 terms 0.27 against 0.62 ms, 2^15 0.35 against 0.53, 2^16 0.52 against
// 0.54; a proof at the 2^15 domain 3.25 against 3.40 ms; at 2^16 a batch of two is SLOWER on this path — a proof 3.76 against 3.56 ms — and a batch
// of four much slower: 4.85).  PLK_MSM_SMALL_MAX overrides (0 = never: A/B knob)
static uint64_t msm_small_max(uint32_t batch) {
    static const long long v = [] { const char *e = getenv("PLK_MSM_SMALL_MAX"); return e ? (long long)strtoull(e, nullptr, 10) : -1ll; }();
    (void)batch;
    return v >= 0 ? (uint64_t)v : (1ull << 15);
}

// -------------------------------------------------------------------------- scalar recoding
// Step 1: every scalar leaves Montgomery form once and is recoded into W signed c-bit digits in
// [-2^(c-1), 2^(c-1)] (c = 17: +-65536), stored as int32 per (commitment, window): digits[(m*W + w)*n + i].
__global__ void __launch_bounds__(MSM_THREADS) msm_digits(ScalarSet set, MsmParams p, int32_t *digits) {
    const uint32_t i = blockIdx.x * MSM_THREADS + threadIdx.x, m = blockIdx.y;
    if (i >= p.n) return;
    Fr k = to_canonical(load_fp(set.v[m] + i));
    uint32_t carry = 0;
    const uint32_t half = 1u << (p.c - 1);
    int32_t *out = digits + (size_t)m * p.windows * p.n + i;
    for (uint32_t w = 0; w < p.windows; w++) {
        uint32_t v = extract_bits(k.l, w * p.c, p.c) + carry;
        int32_t d;
        // the top window is left unsigned: it holds 254 - (W-1)*c bits plus a carry, at most 2^(c-1) — a valid bucket
        if (v >= half && w + 1 < p.windows) { d = (int32_t)v - (int32_t)(1u << p.c); carry = 1; } else { d = (int32_t)v; carry = 0; }
        out[(size_t)w * p.n] = d;
    }
}

// Copy-out of a workgroup's LDS-sorted entries: bin b's run staged[lstart[b] .. lstart[b+1]) goes to entries[gbase[b] ..],
// one run per wave iteration so that the stores are contiguous bursts.  A wave takes 64 consecutive bins: every lane reads
// the three words describing ITS bin once, and the iterations get them by readlane — the first version read them from LDS
// inside the loop, three dependent LDS round trips per 16-entry run (64 iterations per wave: ~9 us of a ~50 us workgroup).
template <int THREADS>
__device__ __forceinline__ void copy_out_runs(const uint32_t *lstart, const uint32_t *gbase, const uint32_t *staged, uint32_t *entries, uint32_t nbins) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t quad = lane >> 4, sub = lane & 15;             // four runs per iteration, 16 lanes each (a run is ~15 entries)
    for (uint32_t b0 = wave * 64; b0 < nbins; b0 += (THREADS / 64) * 64) {
        const uint32_t mine = b0 + lane;
        const uint32_t my_s = mine < nbins ? lstart[mine] : 0, my_e = mine < nbins ? lstart[mine + 1] : 0, my_g = mine < nbins ? gbase[mine] : 0;
        const uint32_t live = nbins - b0 < 64 ? nbins - b0 : 64;
        for (uint32_t j = 0; j < live; j += 4) {
            const int src = (int)(j + quad);                           // (beyond `live`: the lane read zeros, the run is empty)
            const uint32_t s0 = __shfl(my_s, src), len = __shfl(my_e, src) - s0, g0 = __shfl(my_g, src);
            for (uint32_t k = sub; k < len; k += 16) entries[g0 + k] = staged[s0 + k];
        }
    }
}

// Steps 2 and 4: one workgroup per (chunk of DIGIT_CHUNK scalars, global window).  COUNT: coarse-bin
// histogram in LDS -> global.  SCATTER: the chunk's entries are counting-sorted by coarse bin inside LDS,
// one global reservation per (block, bin), then every bin's run is copied out contiguously — full
// 64-256 B bursts instead of 4-byte scattered stores (the first version wrote 13x its payload to HBM).
constexpr int PART_THREADS = 1024;                 // partition workgroups: 16 waves hide the LDS-atomic latency (2^20-term commitment, back to back: 1.53 ms with 256 threads, 1.48 with 512, 1.45 with 1024)
template <bool SCATTER>
__global__ void __launch_bounds__(PART_THREADS) msm_partition(const int32_t *digits, MsmParams p, uint32_t *hist_or_cursor,
                                                              const uint32_t *bin_start, uint32_t *entries) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *lcnt = reinterpret_cast<uint32_t *>(smem);               // [nbins]
    uint32_t *lstart = lcnt + p.nbins;                                 // [nbins + 1]
    uint32_t *gbase = lstart + p.nbins + 1;                            // [nbins]
    uint32_t *staged = gbase + p.nbins;                                // [DIGIT_CHUNK]   (SCATTER only)
    const uint32_t tid = threadIdx.x, gw = blockIdx.y;
    const uint32_t first = blockIdx.x * DIGIT_CHUNK;
    const uint32_t last = first + DIGIT_CHUNK < p.n ? first + DIGIT_CHUNK : p.n;
    const int32_t *dg = digits + (size_t)gw * p.n;
    const uint32_t w = gw % p.windows, set = (gw / p.windows) * p.groups + w % p.groups;    // bucket set of this window
    const uint32_t copy_tag = (w / p.groups) << p.nbits;                                     // which table copy its points come from
    hist_or_cursor += set * p.nbins;
    // the thread's 16 digits are loaded up front, all loads in flight together (one dependent load per loop iteration kept a
    // workgroup resident for ~50 us of pure memory latency — time during which it displaces an accumulate workgroup of the
    // commitment it overlaps), and serve both passes
    constexpr uint32_t PER = DIGIT_CHUNK / PART_THREADS;
    int32_t dreg[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) { const uint32_t i = first + tid + k * PART_THREADS; dreg[k] = i < last ? dg[i] : 0; }
    for (uint32_t b = tid; b < p.nbins; b += PART_THREADS) lcnt[b] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const int32_t d = dreg[k];
        if (d) { uint32_t mg = (uint32_t)(d < 0 ? -d : d) - 1; atomicAdd(&lcnt[mg >> p.fine_bits], 1u); }
    }
    __syncthreads();
    if (!SCATTER) {
        for (uint32_t b = tid; b < p.nbins; b += PART_THREADS) if (lcnt[b]) atomicAdd(&hist_or_cursor[b], lcnt[b]);
        return;
    }
    bin_start += set * p.nbins;
    if (tid < 64) {                                                    // exclusive scan of <= 256 counts by one wave
        uint32_t per = (p.nbins + 63) / 64, lo = tid * per, hi = lo + per < p.nbins ? lo + per : p.nbins, sum = 0;
        for (uint32_t b = lo; b < hi && b < p.nbins; b++) sum += lcnt[b];
        uint32_t v = sum;
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(v, off); if ((int)tid >= off) v += t; }
        uint32_t run = v - sum;
        for (uint32_t b = lo; b < hi && b < p.nbins; b++) { lstart[b] = run; run += lcnt[b]; }
        if (tid == 63) lstart[p.nbins] = v;
    }
    __syncthreads();
    for (uint32_t b = tid; b < p.nbins; b += PART_THREADS) {
        uint32_t cnt = lcnt[b];
        gbase[b] = cnt ? bin_start[b] + atomicAdd(&hist_or_cursor[b], cnt) : 0;
        lcnt[b] = 0;                                                   // reused as the in-bin cursor
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const int32_t d = dreg[k];
        if (d) {
            const uint32_t i = first + tid + k * PART_THREADS;
            uint32_t mg = (uint32_t)(d < 0 ? -d : d) - 1, bin = mg >> p.fine_bits;
            staged[lstart[bin] + atomicAdd(&lcnt[bin], 1u)] = ((copy_tag | i) << 8) | (d < 0 ? 0x80u : 0u) | (mg & ((1u << p.fine_bits) - 1));
        }
    }
    __syncthreads();
    copy_out_runs<PART_THREADS>(lstart, gbase, staged, entries, p.nbins);
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

// ------------------------------------------------------------------ fused recoding (one bucket set per commitment)
// With the table of shifted copies every window of a commitment drops into ONE bucket set (groups == 1: up to 2^20 terms,
// c = 17, 15 windows) — then a scalar's 15 entries can be produced where the scalar is read, and the int32 digit array
// (60 B per term written by msm_digits, read twice by msm_partition) disappears: both passes read the 32-byte scalar,
// leave Montgomery form (one product) and recode in registers.  Per term the pre-phase moves 64 B + the 60 B of entries
// instead of 212 B, in two launches instead of three (0.22 -> ~0.1 ms in front of a 2^20-term commitment: it is on the
// critical path of every round of a proof, and in a stream of commitments its workgroups displace accumulate workgroups).
// The workgroup of the scatter pass takes RC_SCALARS scalars x 15 windows = 15360 entries, the same LDS staging volume
// and the same ~15-entry runs per (workgroup, coarse bin) as the per-window partition it replaces.
constexpr int RC_THREADS = 1024, RC_SCALARS = 1024;                    // (RC_WINDOWS, recode17: msm_shape.h — shared with msm_small.hip)
static_assert(RC_WINDOWS == 254 / 17 + 1, "the fused path is the c = 17 shape");

// true for every lane of the wave iff all its live lanes hold the same scalar (then every window's digit is the same in all of them)
__device__ __forceinline__ bool wave_same_scalar(const Fr &k, bool live, uint64_t live_mask) {
    const int leader = __ffsll((unsigned long long)live_mask) - 1;
    uint32_t diff = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) diff |= k.l[i] ^ (uint32_t)__shfl((int)k.l[i], leader);
    return __ballot(live && diff != 0) == 0;
}

// pass 1: coarse-bin histogram of the commitment's bucket set; RC_COUNT_PER scalars per thread keep the global atomics at
// one per (workgroup, bin) for 4096 scalars
constexpr int RC_COUNT_PER = 4;
__global__ void __launch_bounds__(RC_THREADS) msm_recode_count(ScalarSet set, MsmParams p, uint32_t *hist) {
    __shared__ uint32_t lcnt[1024];
    const uint32_t tid = threadIdx.x, m = blockIdx.y;
    lcnt[tid] = 0;
    __syncthreads();
    Fr k[RC_COUNT_PER];
    bool live[RC_COUNT_PER];
#pragma unroll
    for (int r = 0; r < RC_COUNT_PER; r++) {                      // all loads in flight together
        const uint32_t i = (blockIdx.x * RC_COUNT_PER + r) * RC_THREADS + tid;
        live[r] = i < p.n;
        if (live[r]) k[r] = load_fp(set.v[m] + i);
    }
#pragma unroll
    for (int r = 0; r < RC_COUNT_PER; r++) {
        // (a wave of 64 equal scalars — all-ones, all-(r-1), a constant column — would hit ONE word with every LDS atomic below, 64-way
        //  serialised; such a wave counts once per window instead: wave_same_scalar, ~20 instructions per scalar)
        const uint64_t lv = __ballot(live[r]);
        if (!lv) continue;
        const bool same = wave_same_scalar(k[r], live[r], lv);
        if (!live[r]) continue;
        int32_t d[RC_WINDOWS];
        recode17(to_canonical(k[r]), d);
        if (same) {
            if ((tid & 63) == (uint32_t)(__ffsll((unsigned long long)lv) - 1)) {
#pragma unroll
                for (uint32_t w = 0; w < RC_WINDOWS; w++)
                    if (d[w]) { const uint32_t mg = (uint32_t)(d[w] < 0 ? -d[w] : d[w]) - 1; atomicAdd(&lcnt[mg >> p.fine_bits], (uint32_t)__popcll(lv)); }
            }
            continue;
        }
#pragma unroll
        for (uint32_t w = 0; w < RC_WINDOWS; w++)
            if (d[w]) { const uint32_t mg = (uint32_t)(d[w] < 0 ? -d[w] : d[w]) - 1; atomicAdd(&lcnt[mg >> p.fine_bits], 1u); }
    }
    __syncthreads();
    if (tid < p.nbins && lcnt[tid]) atomicAdd(&hist[m * p.nbins + tid], lcnt[tid]);
}

// pass 2: the workgroup's 15 x 1024 entries are counting-sorted by coarse bin inside LDS, one global reservation per
// (workgroup, bin), contiguous copy-out (same scheme as msm_partition<true>)
__global__ void __launch_bounds__(RC_THREADS) msm_recode_scatter(ScalarSet set, MsmParams p, uint32_t *cursor, const uint32_t *bin_start, uint32_t *entries) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *lcnt = reinterpret_cast<uint32_t *>(smem);               // [nbins]
    uint32_t *lstart = lcnt + p.nbins;                                 // [nbins + 1]
    uint32_t *gbase = lstart + p.nbins + 1;                            // [nbins]
    uint32_t *staged = gbase + p.nbins;                                // [RC_SCALARS * RC_WINDOWS]
    const uint32_t tid = threadIdx.x, m = blockIdx.y;
    const uint32_t i = blockIdx.x * RC_SCALARS + tid;
    cursor += m * p.nbins; bin_start += m * p.nbins;
    int32_t d[RC_WINDOWS];
    const bool live = i < p.n;
    const uint64_t lv = __ballot(live);
    bool same = false;                                                 // every live lane of this wave holds the same scalar (see msm_recode_count)
    uint32_t lane_rank = 0, wave_live = 0;
    if (live) {
        const Fr k = load_fp(set.v[m] + i);
        same = wave_same_scalar(k, true, lv);
        recode17(to_canonical(k), d);
        lane_rank = (uint32_t)__popcll(lv & ((1ull << (tid & 63)) - 1)); wave_live = (uint32_t)__popcll(lv);
    } else {
#pragma unroll
        for (uint32_t w = 0; w < RC_WINDOWS; w++) d[w] = 0;
    }
    for (uint32_t b = tid; b < p.nbins; b += RC_THREADS) lcnt[b] = 0;
    __syncthreads();
    // the counting pass already hands every entry its rank inside (workgroup, bin) — the value the atomic returns — so the staging pass below needs no second
    // atomic per entry (late round 6: 30 LDS atomics per scalar in this kernel -> 15; measured 52.1 -> 50.9 us at 2^20 scalars: the atomics are not what bounds it)
    uint32_t at[RC_WINDOWS];
#pragma unroll
    for (uint32_t w = 0; w < RC_WINDOWS; w++) {
        at[w] = 0;
        if (d[w]) {
            const uint32_t mg = (uint32_t)(d[w] < 0 ? -d[w] : d[w]) - 1;
            if (!same) at[w] = atomicAdd(&lcnt[mg >> p.fine_bits], 1u);
            else {                                                     // one reservation for the wave's run, positions by rank
                uint32_t base = 0;
                if (lane_rank == 0) base = atomicAdd(&lcnt[mg >> p.fine_bits], wave_live);
                at[w] = (uint32_t)__shfl((int)base, __ffsll((unsigned long long)lv) - 1) + lane_rank;
            }
        }
    }
    __syncthreads();
    // exclusive scan of the <= 1024 counts by the whole workgroup (one bin per thread: wave scan + the 16 wave totals), and the
    // global reservation of every bin's run issued right away: its round trip to L2 overlaps the staging pass below, which only
    // needs the LOCAL offsets
    {
        const uint32_t cnt = tid < p.nbins ? lcnt[tid] : 0;
        uint32_t v = cnt;
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(v, off); if ((int)(tid & 63) >= off) v += t; }
        uint32_t *wtot = gbase;                                         // (scratch until the reservations land)
        if ((tid & 63) == 63) wtot[tid >> 6] = v;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < (tid >> 6); w++) before += wtot[w];
        __syncthreads();
        const uint32_t reserved = cnt ? bin_start[tid] + atomicAdd(&cursor[tid], cnt) : 0;    // (tid < nbins whenever cnt != 0)
        if (tid < p.nbins) lstart[tid] = before + v - cnt;
        if (tid == RC_THREADS - 1) lstart[p.nbins] = before + v;
        __syncthreads();
        if (tid < p.nbins) gbase[tid] = reserved;                       // read by the copy-out, after the barrier that follows the staging
    }
    const uint32_t fmask = (1u << p.fine_bits) - 1;
#pragma unroll
    for (uint32_t w = 0; w < RC_WINDOWS; w++) {
        if (!d[w]) continue;
        const uint32_t mg = (uint32_t)(d[w] < 0 ? -d[w] : d[w]) - 1, bin = mg >> p.fine_bits;
        // window w takes its point from copy w of the table (groups == 1): copy_tag = w << nbits
        staged[lstart[bin] + at[w]] = ((((uint32_t)w << p.nbits) | i) << 8) | (d[w] < 0 ? 0x80u : 0u) | (mg & fmask);
    }
    __syncthreads();
    copy_out_runs<RC_THREADS>(lstart, gbase, staged, entries, p.nbins);
}

// --------------------------------------------------------------------- bucket accumulation
// All group arithmetic below runs on the 9 x 29-bit lazy field layer (ec29_dev.h); the resident SRS
// copy it gathers from is kept in that layer's 2^261 Montgomery domain (srs_to_w_kernel).
__device__ __forceinline__ XyzzW shfl_xor_w(const XyzzW &v, int mask) {
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
__device__ __forceinline__ XyzzW shfl_down_w(const XyzzW &v, int delta) {
    XyzzW r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.x.l[i] = __shfl_down(v.x.l[i], delta);
        r.y.l[i] = __shfl_down(v.y.l[i], delta);
        r.zz.l[i] = __shfl_down(v.zz.l[i], delta);
        r.zzz.l[i] = __shfl_down(v.zzz.l[i], delta);
    }
    return r;
}

// full additions / doublings are off the hot path (reduction kernels): one out-of-line copy each.
// (Measured: also routing their products through out-of-line routines makes them 1.6x slower — the
//  save/restore traffic around 14 calls outweighs the smaller instruction footprint.)
__device__ __noinline__ void xyzzw_add_call(XyzzW *a, const XyzzW *b) { XyzzW t = *a; xyzzw_add(t, *b); *a = t; }
__device__ __noinline__ void xyzzw_double_call(XyzzW *a) { XyzzW t = *a; *a = xyzzw_double(t); }
__device__ __forceinline__ void xyzzw_add_nl(XyzzW &a, const XyzzW &b) { xyzzw_add_call(&a, &b); }
__device__ __forceinline__ XyzzW xyzzw_double_nl(const XyzzW &a) { XyzzW t = a; xyzzw_double_call(&t); return t; }

__global__ void __launch_bounds__(256) srs_to_w_kernel(G1Affine *out, const G1Affine *in, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine p = load_affine(in + i);
    G1Affine o;
    o.x = pack<FqParams>(csub_p(w_from_s(unpack<FqW>(p.x))));
    o.y = pack<FqParams>(csub_p(w_from_s(unpack<FqW>(p.y))));
    store_fp(&out[i].x, o.x);
    store_fp(&out[i].y, o.y);
}

// ------------------------------------------------------------- shifted copies of the bases
// The SRS is a fixed base: copy k of the table holds 2^(17k) * P_i (affine, W domain), built once per SRS
// (14 x 17 doublings per point and one batched inversion per copy; 64 MB per copy at 2^20 points).
// With all 15 copies resident a 2^20-term commitment has ONE bucket set instead of 15: the reduction
// kernels and the host Horner shrink accordingly, the accumulate kernel gathers from a 1 GB table in
// HBM instead of a cache-resident 64 MB one (measured: no difference, it is bound by the multiplier).
constexpr uint32_t COPY_SHIFT = 17, MAX_COPIES = 15, NORM_K = 32;    // 15 windows of 17 bits cover the 254-bit scalars

__global__ void __launch_bounds__(256) srs_shift_kernel(const G1Affine *prev, G1Xyzz *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine p = load_affine(prev + i);
    G1Xyzz r = xyzz_identity();
    if (!is_inf(p)) {                                         // COPY_SHIFT doublings on the 29-bit layer, one inlined site
        XyzzW a;
        a.x = csub_p(w_from_s(unpack<FqW>(p.x))); a.y = csub_p(w_from_s(unpack<FqW>(p.y)));
        a.zz = w_one<FqW>(); a.zzz = w_one<FqW>();
        for (uint32_t k = 0; k < COPY_SHIFT; k++) a = xyzzw_double(a);
        r = xyzzw_export(a);
    }
    store_xyzz(out + i, r);
}

// XYZZ -> affine with one field inversion per NORM_K points (Montgomery's trick on ZZZ; 1/Z = ZZ/ZZZ)
__global__ void __launch_bounds__(256) srs_normalise_kernel(const G1Xyzz *in, Fq *prefix, G1Affine *out, uint64_t n) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo = t * NORM_K, hi = lo + NORM_K < n ? lo + NORM_K : n;
    if (lo >= n) return;
    Fq acc = Fq::one();
    for (uint64_t i = lo; i < hi; i++) {
        Fq z = load_fp(&in[i].zzz);
        if (!z.is_zero()) acc = ECM(acc, z);
        store_fp(prefix + i, acc);
    }
    Fq inv_acc = inv(acc);
    for (uint64_t i = hi; i-- > lo;) {
        const G1Xyzz q = load_xyzz(in + i);
        G1Affine o; o.x = Fq::zero(); o.y = Fq::zero();
        if (!q.zzz.is_zero()) {
            const Fq before = i > lo ? load_fp(prefix + i - 1) : Fq::one();
            const Fq zi = ECM(inv_acc, before);               // 1 / ZZZ_i
            inv_acc = ECM(inv_acc, q.zzz);
            const Fq iz = ECM(q.zz, zi), izz = ECS(iz);       // 1 / Z, 1 / ZZ
            o.x = ECM(q.x, izz);
            o.y = ECM(q.y, zi);
        }
        store_fp(&out[i].x, o.x);
        store_fp(&out[i].y, o.y);
    }
}

// All 15 copies or none: a commitment that can only address 5 or 3 of them ((copy, index) must fit the 24-bit
// field of an entry) uses every 3rd or 5th copy, so the full table serves every size.
static uint32_t table_copies_for(uint64_t srs_n) {           // the table may take up to 32 GiB of the 288 GB
    return (uint64_t)MAX_COPIES * srs_n * sizeof(G1Affine) <= (32ull << 30) ? MAX_COPIES : 1;
}

static int32_t ensure_base_table(plk_ctx *ctx, uint32_t copies, hipStream_t stream) {
    if (ctx->srs_w_valid && ctx->srs_w_copies >= copies) return PLK_OK;
    const uint64_t n = ctx->srs_n;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    PLK_TRY(ctx->srs_w.reserve((size_t)copies * n * sizeof(G1Affine)));
    G1Affine *table = ctx->srs_w.as<G1Affine>();
    hipLaunchKernelGGL(srs_to_w_kernel, dim3(blocks), dim3(256), 0, stream, table, reinterpret_cast<const G1Affine *>(ctx->srs), n);
    if (copies > 1) {
        DevBuf xyzz, prefix, aff[2];
        PLK_TRY(xyzz.reserve(n * sizeof(G1Xyzz)));
        PLK_TRY(prefix.reserve(n * sizeof(Fq)));
        PLK_TRY(aff[0].reserve(n * sizeof(G1Affine)));
        PLK_TRY(aff[1].reserve(n * sizeof(G1Affine)));
        const G1Affine *prev = reinterpret_cast<const G1Affine *>(ctx->srs);
        const uint32_t nblocks = (uint32_t)(((n + NORM_K - 1) / NORM_K + 255) / 256);
        for (uint32_t k = 1; k < copies; k++) {
            G1Affine *next = aff[k & 1].as<G1Affine>();
            hipLaunchKernelGGL(srs_shift_kernel, dim3(blocks), dim3(256), 0, stream, prev, xyzz.as<G1Xyzz>(), n);
            hipLaunchKernelGGL(srs_normalise_kernel, dim3(nblocks), dim3(256), 0, stream, (const G1Xyzz *)xyzz.as<G1Xyzz>(), prefix.as<Fq>(), next, n);
            hipLaunchKernelGGL(srs_to_w_kernel, dim3(blocks), dim3(256), 0, stream, table + (size_t)k * n, (const G1Affine *)next, n);
            prev = next;
        }
        hipError_t e = hipStreamSynchronize(stream);
        xyzz.release(); prefix.release(); aff[0].release(); aff[1].release();
        PLK_HIP(e);
    } else PLK_HIP(hipStreamSynchronize(stream));             // the table is read by commitments on OTHER slot streams too: complete before any is enqueued
    PLK_HIP(hipGetLastError());
    ctx->srs_w_valid = true;
    ctx->srs_w_copies = copies;
    return PLK_OK;
}

static uint32_t slots_per_task(uint32_t fb) { return (1u << fb) + 2 * MSM_THREADS; }
static uint32_t meta_per_task(uint32_t fb) { return (1u << fb) + 2; }

// Kernel B0 — folds a bucket that kernel A spread over more than HOT_SPAN lanes (repeated scalars: a witness
// full of 0/1 values) into that bucket's otherwise unused PRIMARY slot, RL lanes per task working together.
// Uniform scalars never take this path; the kernel then only reads the bucket offsets.
// RL = 2^RL_LOG lanes per task, RB = FINE / RL buckets per lane — a template parameter of the two kernels since round 4: 32 lanes (a chain of 24
// dependent additions) when the launch leaves SIMDs idle anyway (one or two commitments: latency is all that counts), 16 lanes (27 % fewer
// lane-additions, a chain of 35) when a batch of three or more commitments puts two waves of these chains on every SIMD and the work decides
// (profiles/r04_msm_reduce_rl_ab.txt: 16 lanes everywhere is +6 % on a single commitment and -1 % on a stream of them).
constexpr uint32_t HOT_SPAN = 8;
__device__ __forceinline__ uint32_t bucket_span(const uint32_t *meta, uint32_t b, uint32_t mu) {
    const uint32_t s0 = meta[b], e0 = meta[b + 1];
    return e0 > s0 ? (e0 - 1) / mu - s0 / mu : 0;
}
// The reduction kernels below are chains of dependent additions run by one or two waves per SIMD: they need few issue slots, but every slot
// they wait for lengthens the tail of a commitment, which is on the critical path of a proof.  When they share a SIMD with throughput
// kernels (the background transforms of the prover, the accumulation of another proof in flight) they ask the arbiter to go first.
__device__ __forceinline__ void latency_chain_priority() { __builtin_amdgcn_s_setprio(3); }

template <uint32_t FB, uint32_t RL_LOG>
__global__ void __launch_bounds__(MSM_THREADS) msm_fold_hot(XyzzW *partials, const uint32_t *task_meta, const uint32_t *task_start, uint32_t total_bins) {
    constexpr uint32_t FINE = Shape<FB>::FINE, SLOT_PRIMARY = Shape<FB>::SLOT_PRIMARY, SLOT_HEAD = Shape<FB>::SLOT_HEAD,
                       SLOTS_PER_TASK = Shape<FB>::SLOTS_PER_TASK, META_PER_TASK = Shape<FB>::META_PER_TASK, RL = 1u << RL_LOG, RB = FINE / RL;
    latency_chain_priority();
    const uint32_t gt = blockIdx.x * MSM_THREADS + threadIdx.x;
    const uint32_t task = gt / RL, sub = gt % RL;
    const bool live = task < task_start[total_bins];
    const uint32_t *meta = task_meta + (size_t)(live ? task : 0) * META_PER_TASK;
    const uint32_t nc_raw = live ? meta[FINE + 1] : 0, nc = nc_raw & ~TASK_OWNED_BIT;
    if (!__any(nc != 0 && !(nc_raw & TASK_OWNED_BIT))) return;      // (owned tasks have one sum per bucket: nothing is spread)
    const uint32_t mu = (nc_raw & TASK_OWNED_BIT) ? 0x7fffffffu : (nc ? (nc + MSM_THREADS - 1) / MSM_THREADS : 1);
    uint32_t hot_mask = 0;                                    // which of this lane's RB buckets are wide
    if (nc) for (uint32_t k = 0; k < RB; k++) if (bucket_span(meta, RB * sub + k, mu) > HOT_SPAN) hot_mask |= 1u << k;
    if (!__any(hot_mask != 0)) return;
    XyzzW *P = partials + (size_t)(live ? task : 0) * SLOTS_PER_TASK;
    for (uint32_t owner = 0; owner < RL; owner++) {
        const uint32_t m = __shfl(hot_mask, owner, RL);        // uniform over the RL lanes of the task
        for (uint32_t k = 0; k < RB; k++) {
            if (!((m >> k) & 1)) continue;
            const uint32_t b = RB * owner + k, tf = meta[b] / mu, tl = (meta[b + 1] - 1) / mu;
            XyzzW acc = xyzzw_identity();
            for (uint32_t t = tf + 1 + sub; t <= tl; t += RL) { XyzzW o = load_xyzzw(P + SLOT_HEAD + t); xyzzw_add_nl(acc, o); }
            for (uint32_t x = 1; x < RL; x <<= 1) { XyzzW o = shfl_xor_w(acc, x); xyzzw_add_nl(acc, o); }
            if (sub == 0) store_xyzzw(P + SLOT_PRIMARY + b, acc);
        }
    }
}

// Kernel B — RL lanes per task, RB = 128/RL buckets per lane: T = sum_f B_f and S = sum_f (f+1) B_f with
// B_f = the bucket's partial sums from kernel A (PRIMARY, or TAIL of the first lane + HEADs of the following
// lanes, or TAIL + the folded PRIMARY of B0).  Running sums inside the lane (X = sum of the buckets seen so
// far, from the top; Y = sum of the X's), then a shuffle suffix scan and a tree across the RL lanes.
//
// The whole kernel is a chain of full additions.  They are all issued from ONE inlined call site inside a
// step loop — every step is "X += O" or "Y += X" with the operand selected beforehand — so the operands live
// in registers: passing two 144-byte points to an out-of-line addition through scratch memory cost more
// L2 write-through traffic than the arithmetic (measured 0.60 ms for this kernel against 0.3 ms of VALU work).
template <uint32_t FB, uint32_t RL_LOG>
__global__ void __launch_bounds__(MSM_THREADS, 2) msm_task_reduce(const XyzzW *partials, const uint32_t *task_meta,
                                                                   const uint32_t *task_start, XyzzW *task_out, uint32_t total_bins, uint32_t early_exit) {
    constexpr uint32_t FINE = Shape<FB>::FINE, SLOT_PRIMARY = Shape<FB>::SLOT_PRIMARY, SLOT_HEAD = Shape<FB>::SLOT_HEAD, SLOT_TAIL = Shape<FB>::SLOT_TAIL,
                       SLOTS_PER_TASK = Shape<FB>::SLOTS_PER_TASK, META_PER_TASK = Shape<FB>::META_PER_TASK, RL = 1u << RL_LOG, RB = FINE / RL, RB_LOG = FB - RL_LOG;
    static_assert(FB >= RL_LOG + 1, "at least two buckets per lane");
    latency_chain_priority();
    const uint32_t gt = blockIdx.x * MSM_THREADS + threadIdx.x;
    const uint32_t task = gt / RL, sub = gt % RL;
    const bool live = task < task_start[total_bins];
    // the grid covers the UPPER BOUND of the task count (bins + entries / TASK_MAX: about twice the tasks of uniform scalars); a wave without a live task would
    // still walk the USTEPS tree steps on identities — and, from three commitments on, share its SIMD with a live wave whose every step it then doubles
    if (early_exit && !__any(live)) return;
    const uint32_t *meta = task_meta + (size_t)(live ? task : 0) * META_PER_TASK;
    const uint32_t nc_raw = live ? meta[FINE + 1] : 0, nc = nc_raw & ~TASK_OWNED_BIT;
    const uint32_t mu = (nc_raw & TASK_OWNED_BIT) ? 0x7fffffffu : (nc ? (nc + MSM_THREADS - 1) / MSM_THREADS : 1);   // (owned: every bucket is whole — PRIMARY)
    const XyzzW *P = partials + (size_t)(live ? task : 0) * SLOTS_PER_TASK;
    XyzzW X = xyzzw_identity(), Y = xyzzw_identity();
    uint32_t k = nc ? RB : 0;                                 // buckets of this lane still to open (top first)
    uint32_t piece = 0, piece_end = 0;                        // slots of the open bucket still to add
    bool pending_sum = false;
    constexpr uint32_t USTEPS = RL_LOG + RB_LOG + 1 + RL_LOG;
    uint32_t ustep = 0;
    for (;;) {
        const bool done = (piece >= piece_end) && !pending_sum && k == 0;
        const bool lockstep = __all(done);                    // every lane of the wave has folded its buckets
        if (lockstep && ustep == USTEPS) break;
        XyzzW O = xyzzw_identity();                           // (a finished lane adds the identity: no-op)
        bool to_y = false;
        if (!lockstep) {
            // (decoding one step ahead to overlap the load with the addition was measured slower: the extra
            //  36 live registers spill)
            if (piece < piece_end) { O = load_xyzzw(P + piece); piece++; }
            else if (pending_sum) { to_y = true; pending_sum = false; }
            else if (k > 0) {
                k--;
                const uint32_t b = RB * sub + k, s0 = meta[b], e0 = meta[b + 1];
                if (e0 > s0) {
                    const uint32_t tf = s0 / mu, tl = (e0 - 1) / mu;
                    pending_sum = true;
                    if (tf == tl) O = load_xyzzw(P + SLOT_PRIMARY + b);
                    else {
                        O = load_xyzzw(P + SLOT_TAIL + tf);
                        if (tl - tf > HOT_SPAN) { piece = SLOT_PRIMARY + b; piece_end = piece + 1; }      // folded by B0
                        else { piece = SLOT_HEAD + tf + 1; piece_end = SLOT_HEAD + tl + 1; }
                    }
                } else to_y = true;                           // empty bucket: only Y += X
            }
        } else {
            // S = sum_sub Y_sub + RB * sum_{sub >= 1} R_sub with R_sub = sum_{s >= sub} X_s (suffix scan); T = R_0
            if (ustep < RL_LOG) {
                const uint32_t off = 1u << ustep;
                O = shfl_down_w(X, off);
                if (sub + off >= RL) O = xyzzw_identity();
            } else {
                if (ustep == RL_LOG) {
                    if (live && sub == 0) store_xyzzw(task_out + 2 * (size_t)task + 1, X);
                    if (sub == 0) X = xyzzw_identity();
                }
                if (ustep < RL_LOG + RB_LOG) O = X;                               // doubling (the addition handles P + P)
                else if (ustep == RL_LOG + RB_LOG) O = Y;
                else O = shfl_xor_w(X, 1 << (ustep - RL_LOG - RB_LOG - 1));
            }
            ustep++;
        }
        if (to_y) O = Y;
        XyzzW Tacc = X;
        xyzzw_add(Tacc, O);                                   // the one addition site of the kernel
        if (to_y) Y = Tacc; else X = Tacc;
    }
    if (live && sub == 0) store_xyzzw(task_out + 2 * (size_t)task, X);
}

// The same reduction by quads of lanes (late round 6; ec29_quad_dev.h, distributed form): RL = 16 QUADS per task (one wave), RB = FINE / 16 buckets per quad,
// every step one four-lane addition (2.6 us for a wave alone on its SIMD against 7.3 for the lane-wise addition).  Launched for ONE commitment (1024 waves:
// one per SIMD; 191 -> 142 us at 2^20 terms); batches keep the lane-wise kernel, whose lanes do the same work in a quarter of the lane-instructions.  The RB_LOG doublings (X + X) leave through the addition's rare branch.
template <uint32_t FB>
__global__ void __launch_bounds__(MSM_THREADS) msm_task_reduce_quad(const XyzzW *partials, const uint32_t *task_meta, const uint32_t *task_start, XyzzW *task_out, uint32_t total_bins, uint32_t early_exit) {
    constexpr uint32_t FINE = Shape<FB>::FINE, SLOT_PRIMARY = Shape<FB>::SLOT_PRIMARY, SLOT_HEAD = Shape<FB>::SLOT_HEAD, SLOT_TAIL = Shape<FB>::SLOT_TAIL,
                       SLOTS_PER_TASK = Shape<FB>::SLOTS_PER_TASK, META_PER_TASK = Shape<FB>::META_PER_TASK, RL_LOG = 4, RL = 1u << RL_LOG, RB = FINE / RL, RB_LOG = FB - RL_LOG;
    latency_chain_priority();
    const uint32_t task = blockIdx.x * (MSM_THREADS / 64) + (threadIdx.x >> 6), sub = (threadIdx.x & 63u) >> 2, coord = threadIdx.x & 3u;
    const bool live = task < task_start[total_bins];
    if (early_exit && !live) return;                          // (wave-uniform: one wave = one task; see msm_task_reduce)
    const uint32_t *meta = task_meta + (size_t)(live ? task : 0) * META_PER_TASK;
    const uint32_t nc_raw = live ? meta[FINE + 1] : 0, nc = nc_raw & ~TASK_OWNED_BIT;
    const uint32_t mu = (nc_raw & TASK_OWNED_BIT) ? 0x7fffffffu : (nc ? (nc + MSM_THREADS - 1) / MSM_THREADS : 1);
    const XyzzW *P = partials + (size_t)(live ? task : 0) * SLOTS_PER_TASK;
    FqW9 X = w_zero<FqW>(), Y = w_zero<FqW>();
    uint32_t k = nc ? RB : 0;                                 // buckets of this quad still to open (top first)
    uint32_t piece = 0, piece_end = 0;                        // slots of the open bucket still to add
    bool pending_sum = false;
    constexpr uint32_t USTEPS = RL_LOG + RB_LOG + 1 + RL_LOG;
    uint32_t ustep = 0;
    for (;;) {
        const bool done = (piece >= piece_end) && !pending_sum && k == 0;
        const bool lockstep = __all(done);                    // every quad of the wave (= the task) has folded its buckets
        if (lockstep && ustep == USTEPS) break;
        FqW9 O = w_zero<FqW>();
        bool to_y = false;
        if (!lockstep) {
            if (piece < piece_end) { O = load_coord(P + piece, coord); piece++; }
            else if (pending_sum) { to_y = true; pending_sum = false; }
            else if (k > 0) {
                k--;
                const uint32_t b = RB * sub + k, s0 = meta[b], e0 = meta[b + 1];
                if (e0 > s0) {
                    const uint32_t tf = s0 / mu, tl = (e0 - 1) / mu;
                    pending_sum = true;
                    if (tf == tl) O = load_coord(P + SLOT_PRIMARY + b, coord);
                    else {
                        O = load_coord(P + SLOT_TAIL + tf, coord);
                        if (tl - tf > HOT_SPAN) { piece = SLOT_PRIMARY + b; piece_end = piece + 1; }      // folded by B0
                        else { piece = SLOT_HEAD + tf + 1; piece_end = SLOT_HEAD + tl + 1; }
                    }
                } else to_y = true;                           // empty bucket: only Y += X
            }
        } else {
            // S = sum_sub Y_sub + RB * sum_{sub >= 1} R_sub with R_sub = sum_{s >= sub} X_s (suffix scan); T = R_0
            if (ustep < RL_LOG) {
                const uint32_t off = 1u << ustep;
#pragma unroll
                for (int i = 0; i < 9; i++) O.l[i] = __shfl_down(X.l[i], 4 * off);
                if (sub + off >= RL) O = w_zero<FqW>();
            } else {
                if (ustep == RL_LOG) {
                    if (live && sub == 0) store_coord(task_out + 2 * (size_t)task + 1, coord, X);
                    if (sub == 0) X = w_zero<FqW>();
                }
                if (ustep < RL_LOG + RB_LOG) O = X;                               // doubling (the addition's rare branch handles P + P)
                else if (ustep == RL_LOG + RB_LOG) O = Y;
                else O = coord_shfl_xor(X, 4 << (ustep - RL_LOG - RB_LOG - 1));
            }
            ustep++;
        }
        if (to_y) O = Y;
        const FqW9 Tacc = xyzzw_add_dist(X, O, coord);        // the one addition site of the kernel
        if (to_y) Y = Tacc; else X = Tacc;
    }
    if (live && sub == 0) store_coord(task_out + 2 * (size_t)task, coord, X);
}

// ------------------------------------------------------------------------ bin folding
// A coarse bin cut into many tasks (repeated scalars: all ones, all r-1, a witness of booleans put 2^20 entries into one
// bucket of every window) would be walked task by task by ONE thread of msm_window_sums — a serial chain of full
// additions (all r-1 at 2^20: 3.4 ms against 2.0 ms for uniform scalars).  The per-task sums of a bin simply add up
// (same bucket range), so one wave per such bin folds them first: lane l takes tasks l, l + 64, .. then a shuffle tree;
// the bin's first task receives (sum S, sum T), the others the identity.  Bins with <= BIN_FOLD_MIN tasks are left alone:
// for uniform scalars the kernel only reads the task offsets.
constexpr uint32_t BIN_FOLD_MIN = 4;
__global__ void __launch_bounds__(MSM_THREADS) msm_bin_fold(XyzzW *task_out, const uint32_t *task_start, uint32_t total_bins) {
    latency_chain_priority();
    const uint32_t bin = blockIdx.x * (MSM_THREADS / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
    if (bin >= total_bins) return;
    const uint32_t t0 = task_start[bin], t1 = task_start[bin + 1];
    if (t1 - t0 <= BIN_FOLD_MIN) return;                       // wave-uniform
    for (uint32_t which = 0; which < 2; which++) {            // 0: S, 1: T
        XyzzW acc = xyzzw_identity();
        for (uint32_t t = t0 + lane; t < t1; t += 64) { XyzzW o = load_xyzzw(task_out + 2 * (size_t)t + which); xyzzw_add_nl(acc, o); }
        for (uint32_t x = 1; x < 64; x <<= 1) { XyzzW o = shfl_xor_w(acc, x); xyzzw_add_nl(acc, o); }
        for (uint32_t t = t0 + lane; t < t1; t += 64) store_xyzzw(task_out + 2 * (size_t)t + which, t == t0 ? acc : xyzzw_identity());
    }
}

// ------------------------------------------------------------------------ window reduction
// W_w = sum_t S_t + 2^FB * sum_c c * D_c,  D_c = sum of T_t over the tasks of coarse bin c.
// 256 threads; thread t serves bins t, t + 256, .. (nbins <= 1024: up to four of them).  With c = 256 u + t:
//     sum_c c * D_c = sum_t t * E_t + 256 * sum_{u >= 1} u * F_u,   E_t = sum_u D_{t + 256 u},   F_u = sum_t D_{t + 256 u},
//     sum_t t * E_t = sum_{b < 8} 2^b * G_b,                          G_b = sum of E_t over the t with bit b set.
// One workgroup per role, side by side, each a plain tree sum (its threads first add up their own bins, then 8 tree steps):
// role 0: sum of the S_t; roles 1..8: G_0..G_7; roles 9..: F_1.. .  Round 1 formed sum_t t * E_t with a suffix scan in one
// workgroup (8 more dependent additions of ~8 us on the tail of every commitment); the weights 2^b, 2^FB, 2^8 and u are a
// few dozen host operations.  Pure chains of full additions from one inlined call site (operands in registers, see
// msm_task_reduce).  Results leave in the library's external form (canonical, R = 2^256).
constexpr uint32_t THREADS_LOG = 8;
static_assert((1u << THREADS_LOG) == MSM_THREADS, "THREADS_LOG");
constexpr uint32_t WS_BIT_ROLES = THREADS_LOG, WS_FIRST_F_ROLE = 1 + WS_BIT_ROLES;
__global__ void __launch_bounds__(MSM_THREADS, 2) msm_window_sums(const XyzzW *task_out, const uint32_t *task_start, G1Xyzz *window_out, uint32_t nbins, uint32_t roles) {
    __shared__ __attribute__((aligned(16))) XyzzW sh[MSM_THREADS];
    latency_chain_priority();
    const uint32_t tid = threadIdx.x, w = blockIdx.x, role = blockIdx.y;
    const uint32_t halves = (nbins + MSM_THREADS - 1) / MSM_THREADS;            // 1 .. 4
    const bool takes_part = role == 0 || role >= WS_FIRST_F_ROLE || ((tid >> (role - 1)) & 1);
    uint32_t u = role >= WS_FIRST_F_ROLE ? role - WS_FIRST_F_ROLE + 1 : 0;
    const uint32_t u_end = !takes_part ? u : (role >= WS_FIRST_F_ROLE ? u + 1 : halves);
    uint32_t t = 0, t_end = 0;
    auto open_bin = [&]() {                                   // next non-empty bin of this thread
        for (; u < u_end; u++) {
            const uint32_t bin = tid + MSM_THREADS * u;
            if (bin >= nbins) continue;
            t = task_start[w * nbins + bin]; t_end = task_start[w * nbins + bin + 1];
            if (t < t_end) return;
        }
        t = t_end = 0;
    };
    open_bin();
    XyzzW X = xyzzw_identity();
    uint32_t ustep = 0;
    for (;;) {
        const bool lockstep = __syncthreads_and(t >= t_end);  // (also the barrier that lets sh be rewritten)
        if (lockstep && ustep == THREADS_LOG) break;
        XyzzW O = xyzzw_identity();
        if (!lockstep) {
            if (t < t_end) {
                O = load_xyzzw(task_out + 2 * (size_t)t + (role ? 1 : 0));
                if (++t == t_end) { u++; open_bin(); }
            }
        } else {
            sh[tid] = X;
            __syncthreads();
            const uint32_t off = (MSM_THREADS / 2) >> ustep;
            if (tid < off) O = sh[tid + off];
            ustep++;
        }
        xyzzw_add(X, O);                                      // the one addition site of the kernel
    }
    if (tid == 0) store_xyzz(window_out + (size_t)roles * w + role, xyzzw_export(X));
}

// The same sums by quads of lanes (late round 6; ec29_quad_dev.h, distributed form: lane r of a quad holds coordinate r of its running sum): a full
// addition is four products deep instead of fourteen — 2.6 us per step for a wave that has its SIMD to itself against 7.3 — and the steps carry no
// workgroup barrier (waves run their own bins, then ONE exchange through LDS).  One workgroup of 64 quads per (bucket set, role); role 0 is split in two
// (bins below / from nbins / 2: the second half is the LAST role, the host adds the two) so that no quad walks more than nbins / 128 bins; the G_b roles
// enumerate their nbins / 2 bins directly.  12 + 6 steps of 2.6 us at 1024 bins against 4 + 8 steps of 8.3 us (msm_window_sums: 100 -> ~45 us on the tail of every
// commitment of >= 2^16 terms).
__global__ void __launch_bounds__(MSM_THREADS) msm_window_sums_quad(const XyzzW *task_out, const uint32_t *task_start, G1Xyzz *window_out, uint32_t nbins, uint32_t roles) {
    __shared__ __attribute__((aligned(16))) uint32_t sh[4][4][9];
    latency_chain_priority();
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, coord = tid & 3, quad = tid >> 2, w = blockIdx.x, role = blockIdx.y;
    const bool second_half = role == roles - 1;
    const uint32_t which = (role == 0 || second_half) ? 0u : 1u;
    const uint32_t count = role >= WS_FIRST_F_ROLE && !second_half ? MSM_THREADS : nbins / 2;
    auto bin_of = [&](uint32_t j) -> uint32_t {
        if (second_half) return nbins / 2 + j;
        if (role == 0) return j;
        if (role >= WS_FIRST_F_ROLE) return (role - WS_FIRST_F_ROLE + 1) * MSM_THREADS + j;
        const uint32_t b = role - 1;                          // the j-th bin index with bit b set
        return ((j >> b) << (b + 1)) | (1u << b) | (j & ((1u << b) - 1));
    };
    uint32_t j = quad, t = 0, t_end = 0;
    auto open_bin = [&]() {                                   // next non-empty bin of this quad
        for (; j < count; j += 64) {
            const uint32_t bin = bin_of(j);
            if (bin >= nbins) continue;
            t = task_start[w * nbins + bin]; t_end = task_start[w * nbins + bin + 1];
            if (t < t_end) { j += 64; return; }
        }
        t = t_end = 0;
    };
    open_bin();
    FqW9 X = w_zero<FqW>();
    uint32_t ustep = 0;
    for (;;) {
        const bool lockstep = __all(t >= t_end);              // every quad of the WAVE has walked its bins
        if (lockstep && ustep == 6) break;
        FqW9 O = w_zero<FqW>();
        if (!lockstep) {
            if (t < t_end) {
                O = load_coord(task_out + 2 * (size_t)t + which, coord);
                if (++t == t_end) open_bin();
            }
        } else {
            if (ustep == 4) {                                 // the four waves' sums change hands through LDS (the one barrier of the kernel)
                if (lane < 4) for (int i = 0; i < 9; i++) sh[wave][coord][i] = X.l[i];
                __syncthreads();
                if ((lane >> 2) < 4) { for (int i = 0; i < 9; i++) X.l[i] = sh[lane >> 2][coord][i]; } else X = w_zero<FqW>();
            }
            O = coord_shfl_xor(X, 4 << (ustep & 3));
            ustep++;
        }
        X = xyzzw_add_dist(X, O, coord);                      // the one addition site of the kernel
    }
    if (tid < 4) {                                            // each lane of the first quad exports its coordinate (canonical, R = 2^256; the identity is all zero)
        const bool inf = quad_flag<2>(w_all_zero(X));
        store_fp(&window_out[(size_t)roles * w + role].x + coord, inf ? Fq::zero() : pack<FqParams>(s_from_w(X)));
    }
}

// ------------------------------------------------------------------- tiny inputs: no buckets
__global__ void __launch_bounds__(MSM_THREADS) msm_naive(const G1Affine *bases, const Fr *scalars, uint32_t n, G1Xyzz *block_out) {
    __shared__ __attribute__((aligned(16))) XyzzW sh[MSM_THREADS];
    const uint32_t tid = threadIdx.x, i = blockIdx.x * MSM_THREADS + tid;
    XyzzW acc = xyzzw_identity();
    if (i < n) {
        Fr k = to_canonical(load_fp(scalars + i));
        G1Affine pt = load_affine(bases + i);
        if (!k.is_zero() && !is_inf(pt)) {
            AffW q; q.x = unpack<FqW>(pt.x); q.y = unpack<FqW>(pt.y);
            for (int bit = 253; bit >= 0; bit--) {
                acc = xyzzw_double_nl(acc);
                if ((k.l[bit >> 5] >> (bit & 31)) & 1) xyzzw_add_mixed(acc, q, false);
            }
        }
    }
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t off = MSM_THREADS / 2; off > 0; off >>= 1) {
        if (tid < off) { XyzzW o = sh[tid + off]; xyzzw_add_nl(acc, o); sh[tid] = acc; }
        __syncthreads();
    }
    if (tid == 0) store_xyzz(block_out + blockIdx.x, xyzzw_export(acc));
}

// ------------------------------------------------------------------------------ host side
// Window widths are chosen among those whose top window still has many bits (254 = 19*13 + 7 = 16*15 + 14 = 15*16 + 14):
// a top window of 1-2 bits would put every term into two or three buckets of a single bin.
static uint32_t pick_window_bits(uint64_t n, bool have_table) {
    // with the table of shifted copies every size is best served by 17-bit windows and one bucket set (measured
    // against 13/15-bit windows without it: 2^12 0.50 vs 0.64 ms, 2^14 0.58 vs 0.95, 2^16 0.62 vs 0.98, 2^18 0.87 vs 1.12)
    if (have_table) return COPY_SHIFT;
    if (n < (1u << 17)) return 13;
    if (n < (1u << 19)) return 15;
    return 17;                                                // 15 windows; 2^16 buckets = 512 coarse bins x 128
}

// Buckets per task.  64 (one task per coarse bin at the 2^20-term / one-bucket-set shape, half the bucket-reduction work)
// whenever the coarse part then still fits the 1024 bins msm_window_sums serves; PLK_MSM_FINE_BITS overrides (A/B runs).
static uint32_t pick_fine_bits(uint64_t n, uint32_t c) {
    static const int probe = [] { const char *e = getenv("PLK_MSM_FINE_BITS"); return e ? atoi(e) : 0; }();
    uint32_t fb = (probe == 6 || probe == 7) ? (uint32_t)probe : 6;
    (void)n;
    if (c - 1 - fb > 10) fb = FINE_BITS_MAX;                  // at most 1024 coarse bins
    return fb;
}

int32_t ensure_pinned(plk_ctx *ctx, size_t bytes);

static int32_t slot_pinned(plk_ctx::MsmSlot &S, size_t bytes) {
    if (bytes <= S.pinned_cap) return PLK_OK;
    if (S.pinned) (void)hipHostFree(S.pinned);
    S.pinned = nullptr; S.pinned_cap = 0;
    PLK_HIP(hipHostMalloc(&S.pinned, bytes, hipHostMallocDefault));
    S.pinned_cap = bytes;
    return PLK_OK;
}

// the 2^20-shaped pipeline (recode / partition, accumulate, bucket reduction) for one batch on the slot's stream; everything it needs is
// in the arguments, so that msm_finish_batch can run it again for a short commitment whose lists overflowed (msm_small.hip)
struct BigArgs { const G1Affine *bases; uint64_t srs_n; uint32_t copies, c_bits, nbits; ScalarSet set; uint32_t batch; uint64_t n; };
static int32_t msm_big_launch(plk_ctx *ctx, plk_ctx::MsmSlot &S, hipStream_t stream, const BigArgs &A) {
    const G1Affine *bases = A.bases;
    const uint32_t batch = A.batch, copies = A.copies;
    const uint64_t n = A.n;
    MsmParams p;
    p.n = (uint32_t)n;
    p.c = A.c_bits;
    p.windows = 254 / p.c + 1;
    p.groups = p.windows / copies;                            // copies > 1 only for c = 17: 15 windows, copies | 15
    p.nbits = A.nbits;
    p.copy_stride = copies > 1 ? (uint32_t)(p.groups * A.srs_n) : 0;
    p.fine_bits = pick_fine_bits(n, p.c);
    p.coarse_bits = p.c - 1 - p.fine_bits;
    p.nbins = 1u << p.coarse_bits;
    p.batch = batch;
    static const uint32_t probe_debug = [] { const char *e = getenv("PLK_MSM_DEBUG"); return e ? (uint32_t)atoi(e) : 0u; }();   // experiments only, read once
    p.debug = probe_debug;
    const ScalarSet &set = A.set;
    const uint32_t total_sets = batch * p.groups, total_bins = total_sets * p.nbins, total_windows = batch * p.windows;
    const uint32_t max_tasks = total_bins + (uint32_t)(((uint64_t)total_windows * n) / TASK_MAX) + 1;
    PLK_TRY(S.a.reserve((size_t)(3 * total_bins + 4) * sizeof(uint32_t)));               // hist/cursor, bin_start, task_start
    PLK_TRY(S.b.reserve((size_t)total_windows * n * sizeof(uint32_t)));                   // entries
    const uint32_t META_PER_TASK = meta_per_task(p.fine_bits), SLOTS_PER_TASK = slots_per_task(p.fine_bits);
    PLK_TRY(S.c.reserve((size_t)max_tasks * 2 * sizeof(XyzzW) + (size_t)max_tasks * META_PER_TASK * 4));  // per-task (S, T) + bucket offsets
    PLK_TRY(S.e.reserve((size_t)max_tasks * SLOTS_PER_TASK * sizeof(XyzzW)));             // lane partial sums
    PLK_TRY(S.d.reserve((size_t)13 * total_sets * sizeof(G1Xyzz)));                    // per bucket set: sum S, G_0..G_7, F_1..F_3
    uint32_t *hist = S.a.as<uint32_t>(), *bin_start = hist + total_bins, *task_start = bin_start + total_bins + 1;
    uint32_t *entries = S.b.as<uint32_t>();
    XyzzW *task_out = S.c.as<XyzzW>();
    uint32_t *task_meta = reinterpret_cast<uint32_t *>(task_out + 2 * (size_t)max_tasks);
    XyzzW *partials = S.e.as<XyzzW>();
    G1Xyzz *window_out = S.d.as<G1Xyzz>();

    PLK_HIP(hipMemsetAsync(hist, 0, total_bins * sizeof(uint32_t), stream));
    static std::atomic<bool> attr_set{false};                 // (several contexts may commit from several host threads)
    if (!attr_set) {
        PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_partition<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_recode_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        PLK_TRY(msm_accumulate_prepare());
        attr_set = true;
    }
    // PLK_MSM_FUSED_RECODE=0 (A/B knob): the three-launch pre-phase through the digit array, as for several bucket sets
    static const bool fused_ok = [] { const char *e = getenv("PLK_MSM_FUSED_RECODE"); return !(e && e[0] == '0'); }();
    if (fused_ok && p.groups == 1 && p.c == 17 && p.windows == RC_WINDOWS && p.nbins <= 1024) {
        // one bucket set per commitment (the 2^20 shape): digits never leave the registers (msm_recode_count / _scatter)
        const uint32_t cblocks = (uint32_t)((n + (uint64_t)RC_THREADS * RC_COUNT_PER - 1) / ((uint64_t)RC_THREADS * RC_COUNT_PER));
        const uint32_t sblocks = (uint32_t)((n + RC_SCALARS - 1) / RC_SCALARS);
        const size_t lds = (size_t)(3 * p.nbins + 1 + RC_SCALARS * RC_WINDOWS) * sizeof(uint32_t);
        hipLaunchKernelGGL(msm_recode_count, dim3(cblocks, batch), dim3(RC_THREADS), 0, stream, set, p, hist);
        hipLaunchKernelGGL(msm_scan_bins, dim3(1), dim3(1024), 0, stream, hist, bin_start, task_start, total_bins);
        hipLaunchKernelGGL(msm_recode_scatter, dim3(sblocks, batch), dim3(RC_THREADS), lds, stream, set, p, hist, (const uint32_t *)bin_start, entries);
    } else {
        PLK_TRY(S.f.reserve((size_t)total_windows * n * sizeof(int32_t)));
        int32_t *digits = S.f.as<int32_t>();
        hipLaunchKernelGGL(msm_digits, dim3((uint32_t)((n + MSM_THREADS - 1) / MSM_THREADS), batch), dim3(MSM_THREADS), 0, stream, set, p, digits);
        const uint32_t pblocks = (uint32_t)((n + DIGIT_CHUNK - 1) / DIGIT_CHUNK);
        const size_t plds_count = (size_t)p.nbins * sizeof(uint32_t);
        const size_t plds_scatter = (size_t)(3 * p.nbins + 1 + DIGIT_CHUNK) * sizeof(uint32_t);
        hipLaunchKernelGGL(msm_partition<false>, dim3(pblocks, total_windows), dim3(PART_THREADS), plds_count, stream, (const int32_t *)digits, p, hist, (const uint32_t *)nullptr, (uint32_t *)nullptr);
        hipLaunchKernelGGL(msm_scan_bins, dim3(1), dim3(1024), 0, stream, hist, bin_start, task_start, total_bins);
        hipLaunchKernelGGL(msm_partition<true>, dim3(pblocks, total_windows), dim3(PART_THREADS), plds_scatter, stream, (const int32_t *)digits, p, hist, (const uint32_t *)bin_start, entries);
    }
    if (ctx->ev_on) PLK_HIP(hipEventRecord(S.ev[0], stream));
    // lanes per task of the bucket reduction (see msm_task_reduce): 16 for a batch of three or more commitments, 32 otherwise; PLK_MSM_RL_LOG=4|5 forces one (A/B runs)
    static const int probe_rl = [] { const char *e = getenv("PLK_MSM_RL_LOG"); return e ? atoi(e) : 0; }();
    // (since late round 6 also 16 when another commitment is in flight on this context: a stream of commitments is work-bound — every reduction wave displaces an
    //  accumulation wave of the next commitment for as long as it lives — and 16 lanes per task are fewer wave-microseconds; PLK_MSM_RL_STREAM=0: 32 as before)
    static const bool rl_stream = [] { const char *e = getenv("PLK_MSM_RL_STREAM"); return !(e && e[0] == '0'); }();
    const bool others_in_flight = ctx->msm_enq != ctx->msm_fin;
    const uint32_t rl_log = (probe_rl == 4 || probe_rl == 5) ? (uint32_t)probe_rl : ((batch >= 3 || (rl_stream && others_in_flight)) ? 4u : 5u);
    const uint32_t rblocks = ((max_tasks << rl_log) + MSM_THREADS - 1) / MSM_THREADS;
    auto launch_shape = [&](auto fb_tag) {
        constexpr uint32_t FB = decltype(fb_tag)::value;
        // PLK_MSM_ONE_WAVE=1 (measurement knob): the same kernel compiled for one wave per SIMD (512 registers, no spill)
        static const bool one_wave = getenv("PLK_MSM_ONE_WAVE") != nullptr;
        // commitments of <= 2^16 terms: the build whose lanes own the buckets of an evenly filled task (msm_accumulate.hip; PLK_MSM_OWNED_MAX overrides, 0 = never)
        static const long long owned_max = [] { const char *e = getenv("PLK_MSM_OWNED_MAX"); return e ? (long long)strtoull(e, nullptr, 10) : (1ll << 16); }();
        const int variant = one_wave ? 1 : (FB == 6 && (long long)n <= owned_max ? 2 : 0);
        msm_accumulate_launch(FB, variant, max_tasks, stream, bases, (const uint32_t *)entries, (const uint32_t *)bin_start, (const uint32_t *)task_start, partials, task_meta, p);
        if (ctx->ev_on) (void)hipEventRecord(S.ev[1], stream);
        (void)hipEventRecord(S.acc_done, stream);
        // ONE commitment: the bucket reduction by quads of lanes (PLK_MSM_TR_QUAD=0: the lane-wise kernel for every batch size, A/B knob).  A batch of two is
        // 2048 waves of it — two per SIMD, every step twice as long: 269 us inside a proof against ~230 lane-wise —
        static const bool tr_quad = [] { const char *e = getenv("PLK_MSM_TR_QUAD"); return !(e && e[0] == '0'); }();
        static const uint32_t early_exit = [] { const char *e = getenv("PLK_MSM_REDUCE_EARLY_EXIT"); return (e && e[0] == '0') ? 0u : 1u; }();   // A/B knob: 0 = dead waves walk the tree steps (rounds 1-6)
        // (only when no other commitment is in flight on this context: the quads do the same additions in 1.5x the lane-instructions, which a stream of
        //  commitments — whose reductions share the GPU with the next accumulation — pays for: three in flight at 2^16 terms 0.243 -> 0.255 ms, measured)
        // and ONE bucket set (above 2^20 terms a commitment has 3 or 5: 5120 tasks are five waves of quads per SIMD — 0.71 ms at 2^22 terms against ~0.45 lane-wise)
        if (tr_quad && probe_rl == 0 && total_sets == 1 && ctx->msm_enq == ctx->msm_fin) {
            hipLaunchKernelGGL((msm_fold_hot<FB, 5>), dim3(rblocks), dim3(MSM_THREADS), 0, stream, partials, (const uint32_t *)task_meta, (const uint32_t *)task_start, total_bins);
            hipLaunchKernelGGL((msm_task_reduce_quad<FB>), dim3((max_tasks + MSM_THREADS / 64 - 1) / (MSM_THREADS / 64)), dim3(MSM_THREADS), 0, stream,
                               (const XyzzW *)partials, (const uint32_t *)task_meta, (const uint32_t *)task_start, task_out, total_bins, early_exit);
        } else if (rl_log == 4) {
            hipLaunchKernelGGL((msm_fold_hot<FB, 4>), dim3(rblocks), dim3(MSM_THREADS), 0, stream, partials, (const uint32_t *)task_meta, (const uint32_t *)task_start, total_bins);
            hipLaunchKernelGGL((msm_task_reduce<FB, 4>), dim3(rblocks), dim3(MSM_THREADS), 0, stream,
                               (const XyzzW *)partials, (const uint32_t *)task_meta, (const uint32_t *)task_start, task_out, total_bins, early_exit);
        } else {
            hipLaunchKernelGGL((msm_fold_hot<FB, 5>), dim3(rblocks), dim3(MSM_THREADS), 0, stream, partials, (const uint32_t *)task_meta, (const uint32_t *)task_start, total_bins);
            hipLaunchKernelGGL((msm_task_reduce<FB, 5>), dim3(rblocks), dim3(MSM_THREADS), 0, stream,
                               (const XyzzW *)partials, (const uint32_t *)task_meta, (const uint32_t *)task_start, task_out, total_bins, early_exit);
        }
    };
    if (p.fine_bits == 6) launch_shape(std::integral_constant<uint32_t, 6>{}); else launch_shape(std::integral_constant<uint32_t, 7>{});
    hipLaunchKernelGGL(msm_bin_fold, dim3((total_bins + MSM_THREADS / 64 - 1) / (MSM_THREADS / 64)), dim3(MSM_THREADS), 0, stream, task_out, (const uint32_t *)task_start, total_bins);
    // points per bucket set left for the host: S (bins below nbins / 2), G_0..G_7, F_1 .., S (bins from nbins / 2)
    const uint32_t roles = WS_FIRST_F_ROLE + (p.nbins + MSM_THREADS - 1) / MSM_THREADS - 1 + 1;
    static const bool ws_quad = [] { const char *e = getenv("PLK_MSM_WS_QUAD"); return !(e && e[0] == '0'); }();      // A/B knob: 0 = the lane-wise kernel (its last role is the identity)
    if (ws_quad) hipLaunchKernelGGL(msm_window_sums_quad, dim3(total_sets, roles), dim3(MSM_THREADS), 0, stream, (const XyzzW *)task_out, (const uint32_t *)task_start, window_out, p.nbins, roles);
    else {
        PLK_HIP(hipMemsetAsync(window_out, 0, (size_t)roles * total_sets * sizeof(G1Xyzz), stream));
        hipLaunchKernelGGL(msm_window_sums, dim3(total_sets, roles - 1), dim3(MSM_THREADS), 0, stream, (const XyzzW *)task_out, (const uint32_t *)task_start, window_out, p.nbins, roles);
    }
    PLK_HIP(hipGetLastError());
    PLK_TRY(slot_pinned(S, (size_t)roles * total_sets * sizeof(G1Xyzz)));
    PLK_HIP(hipMemcpyAsync(S.pinned, window_out, (size_t)roles * total_sets * sizeof(G1Xyzz), hipMemcpyDeviceToHost, stream));
    S.roles = roles;
    S.windows = p.groups;
    S.c_bits = p.c;
    S.fine_bits = p.fine_bits;
    return PLK_OK;
}

// fewer than 4096 terms without the table's copies: one double-and-add per term (msm_naive), block sums to the host
static int32_t msm_naive_launch(plk_ctx::MsmSlot &S, hipStream_t stream, const BigArgs &A) {
    const uint32_t blocks = (uint32_t)((A.n + MSM_THREADS - 1) / MSM_THREADS);
    PLK_TRY(S.d.reserve((size_t)A.batch * blocks * sizeof(G1Xyzz)));
    for (uint32_t m = 0; m < A.batch; m++)
        hipLaunchKernelGGL(msm_naive, dim3(blocks), dim3(MSM_THREADS), 0, stream, A.bases, A.set.v[m], (uint32_t)A.n, S.d.as<G1Xyzz>() + (size_t)m * blocks);
    PLK_HIP(hipGetLastError());
    (void)hipEventRecord(S.acc_done, stream);
    S.pending_parts = blocks;
    S.windows = 0;
    S.c_bits = 0;
    PLK_TRY(slot_pinned(S, (size_t)A.batch * blocks * sizeof(G1Xyzz)));
    PLK_HIP(hipMemcpyAsync(S.pinned, S.d.p, (size_t)A.batch * blocks * sizeof(G1Xyzz), hipMemcpyDeviceToHost, stream));
    return PLK_OK;
}

// `caller` is the stream on which the scalars were produced; the commitment runs on its slot's own stream after an
// event recorded there.  The caller must leave the scalars alone until the matching msm_finish_batch.
int32_t msm_enqueue_batch(plk_ctx *ctx, const Fr *const *scalars_dev, uint32_t batch, uint64_t n, uint64_t base_offset, hipStream_t caller) {
    if (ctx->msm_enq - ctx->msm_fin >= plk_ctx::MSM_SLOTS) { set_error("msm: " + std::to_string(plk_ctx::MSM_SLOTS) + " commitments are already in flight (call the finish function first)"); return PLK_ERR_ARG; }
    uint32_t slot_index = 0;
    while (ctx->slot[slot_index].busy) slot_index++;          // lowest free slot (there is one: fewer than MSM_SLOTS are in flight)
    plk_ctx::MsmSlot &S = ctx->slot[slot_index];
    auto in_flight = [&]() { ctx->fifo[ctx->msm_enq % plk_ctx::MSM_SLOTS] = (uint8_t)slot_index; S.busy = true; ctx->msm_enq++; };
    if (!S.stream) {
        PLK_HIP(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
        PLK_HIP(hipEventCreateWithFlags(&S.ready, hipEventDisableTiming));
        PLK_HIP(hipEventCreateWithFlags(&S.acc_done, hipEventDisableTiming));
    }
    if (ctx->ev_on && !S.ev[0]) { PLK_HIP(hipEventCreate(&S.ev[0])); PLK_HIP(hipEventCreate(&S.ev[1])); }
    PLK_HIP(hipEventRecord(S.ready, caller));
    PLK_HIP(hipStreamWaitEvent(S.stream, S.ready, 0));
    hipStream_t stream = S.stream;
    if (!ctx->srs) { set_error("msm: no SRS uploaded (plk_srs_upload)"); return PLK_ERR_SRS; }
    if (base_offset + n > ctx->srs_n) { set_error("msm: SRS too small for this commitment"); return PLK_ERR_SRS; }
    if (n >= (1ull << 24) + 1) { set_error("msm: more than 2^24 terms in one pass (plk_msm_g1 / plk_msm_g1_dev / plk_prove split longer commitments; the enqueue and batch entry points do not)"); return PLK_ERR_SIZE; }
    if (batch < 1 || batch > MSM_MAX_BATCH) { set_error("msm: batch must be 1..8"); return PLK_ERR_ARG; }
    // resident table of the SRS in the 2^261 domain of the lazy field layer; large commitments use its shifted copies
    uint32_t nbits = 1;
    while ((1ull << nbits) < n) nbits++;
    // short commitments (msm_small.hip) need all 15 copies of the table: every commitment of >= 4096 terms builds them anyway; a shorter one
    // only asks for them when the key is small enough for that to be cheap (<= 2^21 points: 80 ms once per key) or has them already
    const bool small_wanted = n >= 1 && n <= msm_small_max(batch) && table_copies_for(ctx->srs_n) == MAX_COPIES &&
                              (n >= 4096 || ctx->srs_n <= (1ull << 21) || (ctx->srs_w_valid && ctx->srs_w_copies == MAX_COPIES));
    const uint32_t c_bits = small_wanted ? COPY_SHIFT : (n >= 4096 ? pick_window_bits(n, table_copies_for(ctx->srs_n) > 1) : 0);
    uint32_t copies = 1;
    if (c_bits == COPY_SHIFT) {
        copies = table_copies_for(ctx->srs_n);
        PLK_TRY(ensure_base_table(ctx, copies, stream));
        while (copies > 1 && (MAX_COPIES % copies != 0 || ((uint64_t)copies << nbits) > (1ull << 24))) copies--;   // a divisor of 15 that fits the 24-bit (copy, index) field of an entry
        static const int probe_copies = [] { const char *e = getenv("PLK_MSM_COPIES"); return e ? atoi(e) : 0; }();   // tuning probe, read once
        if (probe_copies >= 1 && (uint32_t)probe_copies <= copies && MAX_COPIES % probe_copies == 0) copies = (uint32_t)probe_copies;
    } else PLK_TRY(ensure_base_table(ctx, 1, stream));
    const bool small = small_wanted && copies == MAX_COPIES;
    const G1Affine *bases = ctx->srs_w.as<G1Affine>() + base_offset;
    S.pending_parts = 0;
    S.windows = 0;
    S.batch = batch;
    S.small = false;
    if (n == 0) { (void)hipEventRecord(S.acc_done, stream); in_flight(); return PLK_OK; }
    BigArgs A{bases, ctx->srs_n, copies, c_bits, nbits, ScalarSet{}, batch, n};
    for (uint32_t m = 0; m < batch; m++) A.set.v[m] = scalars_dev[m];
    if (n < 4096 && !small) {
        PLK_TRY(msm_naive_launch(S, stream, A));
        in_flight();
        return PLK_OK;
    }
    // short commitments: three short launches instead of the 2^20-shaped pipeline (msm_small.hip).  They need all 15 table copies.
    if (small) {
        PLK_TRY(slot_pinned(S, (size_t)MSM_MAX_BATCH * SM_PLANES_HOST * sizeof(G1Xyzz) + 16));
        PLK_TRY(msm_small_launch(S, stream, bases, (uint32_t)ctx->srs_n, A.set, batch, (uint32_t)n, ctx->ev_on, S.pinned));
        S.small = true;
        S.fb_bases = bases; S.fb_srs_n = ctx->srs_n; S.fb_n = n; S.fb_copies = copies; S.fb_cbits = c_bits; S.fb_nbits = nbits;
        for (uint32_t m = 0; m < batch; m++) S.fb_scalars[m] = scalars_dev[m];
        in_flight();
        return PLK_OK;
    }
    PLK_TRY(msm_big_launch(ctx, S, stream, A));
    in_flight();
    return PLK_OK;
}

int32_t msm_enqueue(plk_ctx *ctx, const Fr *scalars_dev, uint64_t n, uint64_t base_offset, hipStream_t stream) {
    return msm_enqueue_batch(ctx, &scalars_dev, 1, n, base_offset, stream);
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
int32_t msm_finish_batch(plk_ctx *ctx, hipStream_t, host::HJac *out) {
    using namespace host;
    if (ctx->msm_fin == ctx->msm_enq) { set_error("msm: nothing in flight"); return PLK_ERR_ARG; }
    plk_ctx::MsmSlot &S = ctx->front_slot();
    ctx->last_slot = ctx->fifo[ctx->msm_fin % plk_ctx::MSM_SLOTS];
    ctx->msm_fin++;
    S.busy = false;
    PLK_HIP(hipStreamSynchronize(S.stream));
    if (S.small) {
        const uint32_t flag = *reinterpret_cast<const volatile uint32_t *>(static_cast<const char *>(S.pinned) + (size_t)S.batch * SM_PLANES_HOST * sizeof(G1Xyzz));
        if (flag) {                                           // a bucket list overflowed (msm_small.hip): the ordinary pipeline on the same inputs
            BigArgs A{static_cast<const G1Affine *>(S.fb_bases), S.fb_srs_n, S.fb_copies, S.fb_cbits, S.fb_nbits, ScalarSet{}, S.batch, S.fb_n};
            for (uint32_t m = 0; m < S.batch; m++) A.set.v[m] = static_cast<const Fr *>(S.fb_scalars[m]);
            S.small = false;
            if (S.fb_n < 4096) PLK_TRY(msm_naive_launch(S, S.stream, A)); else PLK_TRY(msm_big_launch(ctx, S, S.stream, A));
            PLK_HIP(hipStreamSynchronize(S.stream));
        }
    }
    for (uint32_t m = 0; m < S.batch; m++) {
        HJac acc = HJac::inf();
        if (S.small) {
            // seventeen plane sums Q_0..Q_16 per commitment: sum_b 2^b * Q_b (bits 0-7: the lo digit, 8-16: the hi digit)
            const uint64_t *raw = reinterpret_cast<const uint64_t *>(S.pinned) + (size_t)16 * SM_PLANES_HOST * m;
            for (int b = (int)SM_PLANES_HOST - 1; b >= 0; b--) acc = jac_add(jac_double(acc), xyzz_host_to_jac(raw + 16 * b));
        } else if (S.windows) {
            // per bucket set the device leaves (sum S over the lower bins, G_0..G_7, F_1, .., F_{halves-1}, sum S over the upper bins);
            // W = sum S + 2^FB * (sum_b 2^b G_b + 2^8 * sum_u u*F_u)
            const size_t per = (size_t)16 * S.roles;
            const uint64_t *raw = reinterpret_cast<const uint64_t *>(S.pinned) + per * m * S.windows;
            for (int w = (int)S.windows - 1; w >= 0; w--) {
                for (uint32_t i = 0; i < S.c_bits; i++) acc = jac_double(acc);
                HJac run = HJac::inf(), d = HJac::inf();                  // sum_u u*F_u = sum of the suffix sums of F
                for (uint32_t r = S.roles - 2; r >= WS_FIRST_F_ROLE; r--) { run = jac_add(run, xyzz_host_to_jac(raw + per * w + 16 * r)); d = jac_add(d, run); }
                for (int b = (int)WS_BIT_ROLES - 1; b >= 0; b--) d = jac_add(jac_double(d), xyzz_host_to_jac(raw + per * w + 16 * (1 + b)));   // Horner over the bit sums
                for (uint32_t i = 0; i < S.fine_bits; i++) d = jac_double(d);
                acc = jac_add(acc, jac_add(jac_add(xyzz_host_to_jac(raw + per * w), xyzz_host_to_jac(raw + per * w + 16 * (S.roles - 1))), d));   // (sum S in two halves)
            }
        } else {
            const uint64_t *raw = reinterpret_cast<const uint64_t *>(S.pinned) + (size_t)16 * m * S.pending_parts;
            for (uint32_t i = 0; i < S.pending_parts; i++) acc = jac_add(acc, xyzz_host_to_jac(raw + 16 * i));
        }
        out[m] = acc;
    }
    return PLK_OK;
}

int32_t msm_finish(plk_ctx *ctx, hipStream_t stream, host::HJac *out) {
    if (ctx->msm_fin != ctx->msm_enq && ctx->front_slot().batch != 1) { set_error("msm_finish: a batch is pending"); return PLK_ERR_ARG; }
    return msm_finish_batch(ctx, stream, out);
}

// From 2^23 terms on, ONE commitment is better run as 2^20-term pieces over successive SRS ranges, three pieces in flight: the
// digit / partition kernels and the bucket reduction of a piece then overlap the accumulation of its neighbours, which a single
// pass cannot do with itself (2^24 terms: 21.8 ms against 24.0 in one pass; at 2^22 the pieces do not pay: 6.5 against 6.15 ms,
// profiles/r02_msm_three_in_flight_ab.txt).  Only with the table of shifted copies: an SRS of more than 2^25 points has none
// (it would exceed 32 GiB), a 2^20-term piece then runs 19 windows instead of 15 and a 2^26-gate proof got 40 % SLOWER that way.
uint64_t msm_pipelined_piece(const plk_ctx *ctx, uint64_t terms) {
    return (terms >= (1ull << 23) && table_copies_for(ctx->srs_n) > 1) ? (1ull << 20) : 0;
}

}  // namespace plk

using namespace plk;

extern "C" {

int32_t plk_msm_g1_enqueue_dev(plk_ctx *ctx, const void *scalars_dev, uint64_t n, uint64_t base_offset, void *stream) {
    if (!ctx || (!scalars_dev && n)) { set_error("plk_msm_g1: bad argument"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    return msm_enqueue(ctx, (const Fr *)scalars_dev, n, base_offset, stream ? (hipStream_t)stream : ctx->stream);
}

// the same FIFO for a batch of `count` (<= 8) commitments of equal length against the same bases: one pass of the kernels
int32_t plk_msm_g1_enqueue_batch_dev(plk_ctx *ctx, const void *const *scalars_dev, uint32_t count, uint64_t n, uint64_t base_offset, void *stream) {
    if (!ctx || !scalars_dev || count == 0 || count > MSM_MAX_BATCH) { set_error("plk_msm_g1_enqueue_batch_dev: bad argument"); return PLK_ERR_ARG; }
    for (uint32_t k = 0; k < count; k++) if (!scalars_dev[k] && n) { set_error("plk_msm_g1_enqueue_batch_dev: null vector"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    return msm_enqueue_batch(ctx, reinterpret_cast<const Fr *const *>(scalars_dev), count, n, base_offset, stream ? (hipStream_t)stream : ctx->stream);
}
static int32_t finish_batch_checked(plk_ctx *ctx, uint32_t count, host::HJac *j) {
    if (ctx->msm_fin == ctx->msm_enq) { set_error("msm: nothing in flight"); return PLK_ERR_ARG; }
    if (ctx->front_slot().batch != count) { set_error("plk_msm_g1_finish_batch: the commitment in flight holds a different batch size"); return PLK_ERR_ARG; }
    return msm_finish_batch(ctx, nullptr, j);
}
int32_t plk_msm_g1_finish_batch(plk_ctx *ctx, plk_g1_jacobian *out, uint32_t count) {
    if (!ctx || !out || count == 0 || count > MSM_MAX_BATCH) { set_error("plk_msm_g1_finish_batch: bad argument"); return PLK_ERR_ARG; }
    host::HJac j[MSM_MAX_BATCH];
    PLK_TRY(finish_batch_checked(ctx, count, j));
    for (uint32_t k = 0; k < count; k++) { memcpy(out[k].x, j[k].x.l, 32); memcpy(out[k].y, j[k].y.l, 32); memcpy(out[k].z, j[k].z.l, 32); }
    return PLK_OK;
}
// The commitment-level sharded entry points assume that EVERY rank calls them with its slice of the scalars (replicate mode).  In owner-computes
// mode the workers sit in plk_comm_serve waiting for a batch header, so an exchange started here would never be answered: refuse instead of hanging.
static int32_t sharded_entry_is_replicate_only(plk_ctx *ctx, const char *who) {
    if (!comm_scatter_owner(ctx) && !comm_scatter_worker(ctx)) return PLK_OK;
    set_error(std::string(who) + ": the communicator is in owner-computes mode (PLK_SHARD_SCATTER) — commitment-level sharded calls need replicate mode");
    return PLK_ERR_ARG;
}
// finish + the context's combiner (one exchange for the whole batch), affine results
int32_t plk_msm_g1_finish_batch_sharded(plk_ctx *ctx, plk_g1_affine *out, uint32_t count) {
    if (!ctx || !out || count == 0 || count > MSM_MAX_BATCH) { set_error("plk_msm_g1_finish_batch_sharded: bad argument"); return PLK_ERR_ARG; }
    PLK_TRY(sharded_entry_is_replicate_only(ctx, "plk_msm_g1_finish_batch_sharded"));
    plk_g1_jacobian j[MSM_MAX_BATCH];
    PLK_TRY(plk_msm_g1_finish_batch(ctx, j, count));
    if (ctx->combine) {
        const int32_t rc = ctx->combine(ctx->combine_user, j, count);
        if (rc != PLK_OK) { set_error("commitment combiner failed"); return rc; }
    }
    for (uint32_t k = 0; k < count; k++) PLK_TRY(plk_g1_sum_jacobian(&j[k], 1, &out[k]));
    return PLK_OK;
}

int32_t plk_msm_g1_finish(plk_ctx *ctx, plk_g1_jacobian *out) {
    if (!ctx || !out) { set_error("plk_msm_g1_finish: bad argument"); return PLK_ERR_ARG; }
    host::HJac j;
    PLK_TRY(msm_finish(ctx, nullptr, &j));
    memcpy(out->x, j.x.l, 32); memcpy(out->y, j.y.l, 32); memcpy(out->z, j.z.l, 32);
    return PLK_OK;
}

// finish + the context's combiner (plk_comm_init / plk_set_commit_shard): the commitment over ALL ranks' shards, affine.
// The exchange of commitment k runs while the kernels of commitment k + 1 (enqueued before this call) occupy the GPU.
int32_t plk_msm_g1_finish_sharded(plk_ctx *ctx, plk_g1_affine *out) {
    if (!ctx || !out) { set_error("plk_msm_g1_finish_sharded: bad argument"); return PLK_ERR_ARG; }
    PLK_TRY(sharded_entry_is_replicate_only(ctx, "plk_msm_g1_finish_sharded"));
    plk_g1_jacobian j;
    PLK_TRY(plk_msm_g1_finish(ctx, &j));
    if (ctx->combine) {
        const int32_t rc = ctx->combine(ctx->combine_user, &j, 1);
        if (rc != PLK_OK) { set_error("commitment combiner failed"); return rc; }
    }
    return plk_g1_sum_jacobian(&j, 1, out);
}

int32_t plk_msm_g1_partial_dev(plk_ctx *ctx, const void *scalars_dev, uint64_t n, uint64_t base_offset, plk_g1_jacobian *out, void *stream) {
    constexpr uint64_t PIECE = 1ull << 24;                    // one pass of the kernels takes at most 2^24 terms
    // long commitments as short pieces, several in flight (msm_pipelined_piece); the FIFO must be empty for that (it is, unless
    // the caller keeps commitments in flight)
    const uint64_t pipe_piece = ctx ? msm_pipelined_piece(ctx, n) : 0;
    const bool pipelined = pipe_piece != 0 && ctx->msm_enq == ctx->msm_fin;
    if (n <= PIECE && !pipelined) {
        PLK_TRY(plk_msm_g1_enqueue_dev(ctx, scalars_dev, n, base_offset, stream));
        return plk_msm_g1_finish(ctx, out);
    }
    if (!ctx || !scalars_dev || !out) { set_error("plk_msm_g1: bad argument"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    host::HJac acc = host::HJac::inf();
    const uint64_t piece = pipelined ? pipe_piece : PIECE;
    const uint32_t depth = pipelined ? plk_ctx::MSM_SLOTS : 1;
    uint64_t off = 0;
    uint32_t inflight = 0;
    int32_t rc = PLK_OK;
    while (off < n || inflight) {                             // successive SRS ranges, summed on the host
        while (rc == PLK_OK && off < n && inflight < depth) {
            rc = msm_enqueue(ctx, (const Fr *)scalars_dev + off, n - off < piece ? n - off : piece, base_offset + off, stream ? (hipStream_t)stream : ctx->stream);
            if (rc == PLK_OK) { off += piece; inflight++; }
        }
        if (rc != PLK_OK) off = n;                            // stop enqueuing, drain what is in flight
        if (!inflight) break;
        host::HJac j;
        const int32_t rf = msm_finish(ctx, nullptr, &j);
        inflight--;
        if (rf != PLK_OK && rc == PLK_OK) rc = rf;
        if (rc == PLK_OK) acc = host::jac_add(acc, j);
    }
    if (rc != PLK_OK) return rc;
    memcpy(out->x, acc.x.l, 32); memcpy(out->y, acc.y.l, 32); memcpy(out->z, acc.z.l, 32);
    return PLK_OK;
}

int32_t plk_msm_g1_dev(plk_ctx *ctx, const void *scalars_dev, uint64_t n, uint64_t base_offset, plk_g1_affine *out, void *stream) {
    if (!out) { set_error("plk_msm_g1: null out"); return PLK_ERR_ARG; }
    plk_g1_jacobian j;
    PLK_TRY(plk_msm_g1_partial_dev(ctx, scalars_dev, n, base_offset, &j, stream));
    return plk_g1_sum_jacobian(&j, 1, out);
}

int32_t plk_msm_g1_batch_dev(plk_ctx *ctx, const void *const *scalars_dev, uint32_t count, uint64_t n, uint64_t base_offset, plk_g1_affine *out, void *stream) {
    if (!ctx || !scalars_dev || !out || count == 0) { set_error("plk_msm_g1_batch_dev: bad argument"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    for (uint32_t done = 0; done < count;) {
        uint32_t b = count - done > MSM_MAX_BATCH ? MSM_MAX_BATCH : count - done;
        PLK_TRY(msm_enqueue_batch(ctx, reinterpret_cast<const Fr *const *>(scalars_dev + done), b, n, base_offset, st));
        host::HJac j[MSM_MAX_BATCH];
        PLK_TRY(msm_finish_batch(ctx, st, j));
        for (uint32_t k = 0; k < b; k++) {
            host::HAffine a = host::jac_to_affine(j[k]);
            memcpy(out[done + k].x, a.x.l, 32); memcpy(out[done + k].y, a.y.l, 32);
        }
        done += b;
    }
    return PLK_OK;
}

int32_t plk_msm_g1(plk_ctx *ctx, const plk_fr *scalars, uint64_t n, uint64_t base_offset, plk_g1_affine *out) {
    if (!ctx || (!scalars && n) || !out) { set_error("plk_msm_g1: bad argument"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(ctx->stage.reserve(n * sizeof(plk_fr) + 32));
    PLK_HIP(hipMemcpyAsync(ctx->stage.p, scalars, n * sizeof(plk_fr), hipMemcpyHostToDevice, ctx->stream));
    return plk_msm_g1_dev(ctx, ctx->stage.p, n, base_offset, out, nullptr);
}

int32_t plk_srs_precompute(plk_ctx *ctx) {
    if (!ctx) { set_error("plk_srs_precompute: null ctx"); return PLK_ERR_ARG; }
    if (!ctx->srs) { set_error("plk_srs_precompute: no SRS resident"); return PLK_ERR_SRS; }
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(ensure_base_table(ctx, table_copies_for(ctx->srs_n), ctx->stream));
    if (ctx->lag.pts) {
        SrsSlotSwap active(ctx, true);
        PLK_TRY(ensure_base_table(ctx, table_copies_for(ctx->srs_n), ctx->stream));
    }
    return PLK_OK;
}

// Two (or more) contexts proving side by side on ONE GPU — the latency-bound ends of one proof's commitments and its strict
// challenge chain leave SIMD time that a second proof fills — need one resident key between them, not one each: `dst` borrows
// `src`'s monomial key, its Lagrange-form key if there is one, and the MSM fixed-base tables of both (0.94 GiB and 40 ms per key
// at 2^20 points), built here if `src` has not built them yet.  Read-only from then on, so commitments of both contexts may run
// concurrently.  `src` keeps ownership: it refuses to replace a key while a borrower exists and must outlive its borrowers' use.
int32_t plk_ctx_share_srs(plk_ctx *dst, plk_ctx *src) {
    if (!dst || !src || dst == src) { set_error("plk_ctx_share_srs: bad argument"); return PLK_ERR_ARG; }
    if (dst->device != src->device) { set_error("plk_ctx_share_srs: the two contexts are on different devices"); return PLK_ERR_ARG; }
    if (src->srs_lender) { set_error("plk_ctx_share_srs: the source context itself borrows its key (share from the owner)"); return PLK_ERR_ARG; }
    if (!src->srs) { set_error("plk_ctx_share_srs: the source context has no key resident"); return PLK_ERR_SRS; }
    if (dst->msm_enq != dst->msm_fin) { set_error("plk_ctx_share_srs: a commitment is still in flight on the destination context"); return PLK_ERR_ARG; }
    if (dst->srs_borrowers.load() > 0) { set_error("plk_ctx_share_srs: the destination context lends its own key to other contexts"); return PLK_ERR_ARG; }
    PLK_TRY(plk_srs_precompute(src));                        // the fallible step first (synchronises src->stream: the tables are complete before anyone
                                                             // reads them) — dst keeps what it has if this fails
    PLK_TRY(srs_replace_guard(dst, "plk_ctx_share_srs"));    // drops a loan dst may hold
    dst->srs_own.release(); dst->lag.own.release();
    dst->lag.w.release();
    srs_make_loan(dst, src);
    return PLK_OK;
}

int32_t plk_set_kernel_timing(plk_ctx *ctx, int32_t on) {
    if (!ctx) { set_error("null ctx"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    ctx->ev_on = on != 0;
    return PLK_OK;
}

int32_t plk_msm_last_kernel_ms(plk_ctx *ctx, float *accumulate_ms) {
    if (!ctx || !accumulate_ms) { set_error("plk_msm_last_kernel_ms: bad argument"); return PLK_ERR_ARG; }
    if (!ctx->ev_on || ctx->msm_fin == 0) { set_error("kernel timing is off (plk_set_kernel_timing) or no commitment finished yet"); return PLK_ERR_ARG; }
    plk_ctx::MsmSlot &S = ctx->slot[ctx->last_slot];                          // the commitment finished last
    if (!S.ev[0]) { set_error("the last commitment was enqueued with kernel timing off"); return PLK_ERR_ARG; }
    PLK_HIP(hipEventSynchronize(S.ev[1]));
    PLK_HIP(hipEventElapsedTime(accumulate_ms, S.ev[0], S.ev[1]));
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
