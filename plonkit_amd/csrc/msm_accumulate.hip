// Kernel A of the MSM — bucket accumulation, the dominant kernel of the library (62 % of the GPU time of a 2^20 proof).
// A translation unit of its own since round 4: compiled with `-mllvm -amdgpu-sched-strategy=max-ilp` (plonkit_amd/build.py).
// Same-box A/B of compiler settings on this kernel (profiles/r04_msm_accumulate_sched_ab.txt): the default scheduler 1.101 ms,
// max-ilp 1.078 (-2.2 %), no post-RA scheduling 1.084, -O2 1.088, pre-RA scheduling off 1.093, max-occupancy 1.097 — and the
// same flag on the WHOLE library made a proof 0.3 ms slower (the latency-bound reduction chains and point-wise kernels lose),
// hence one file.  This is what is left of the "hand-scheduled body" idea: the kernel's time is its VALU instruction count
// (SQ counters: the SIMDs issue ~100 % of the time, the s_nop wait states of the product chains are hidden by the second wave),
// so a schedule can only win what a better instruction ORDER wins — about 2 %.
#include "msm_shape.h"
#include "ec29_quad_dev.h"
#include <atomic>

namespace plk {

// Kernel A — one workgroup per task = one slice (<= CHUNK entries) of a (window, coarse bin): 128 buckets.
//  1. counting sort of the slice by fine bucket inside LDS
//  2. the sorted slice is cut into 256 equal pieces, one per lane: every lane performs the same number of
//     mixed additions whatever the bucket populations are (uniform, witness-like or one hot bucket);
//     a lane starts a new accumulator at each bucket boundary inside its piece
// Only mixed additions happen here (10 products each, ~25 KB of code).  The register budget is the full 256 (two waves per
// SIMD: the second wave hides the wait states of the product-scanning chains; 128 VGPRs = 25 % slower, tools/ubench_w).
// The HOT path (sort, flat loop, mixed addition with distinct x) spills nothing; the code object does report spills —
// msm_accumulate<6,2>: VGPRs 256, "VGPRs Spill" 323, scratch 652 B per lane (round 2: 72 / 176 B) — all of them in the COLD
// branches (P == +-Q: doubling / identity, and the generic addition behind the zero filter's false positives), which round 3's
// lockstep products pushed out of registers.  Measured harmless for uniform scalars; all-(r-1) scalars, where every addition
// of a window meets equal points, run 1.27x the uniform time.  Kernel B folds the partial sums.
//
// OWNED (round 6, the instantiation launched for commitments of <= 2^16 terms; the 2^20 kernel is compiled without it and is the same code as before):
// with few entries per bucket the equal pieces make every bucket a TAIL and several HEADs — at 2^16 terms a bucket's ~15 entries lie in four lanes' pieces of
// four, and msm_task_reduce then walks ~5 partial sums per bucket instead of one (a chain of ~44 dependent full additions instead of 19: 0.33 ms of the 0.79 ms
// of a batch of four commitments, beside an accumulation of 0.24).  A task whose fullest bucket holds <= 8 mu + 24 entries (mu = entries per lane: every task of uniform scalars — the reduction kernels take as long as
// their slowest wave, so ALL tasks have to qualify for them to gain) is
// accumulated the way msm_small.hip does it instead: the four lanes 4 b .. 4 b + 3 OWN bucket b, lane j takes entries j, j + 4, .. of its run, the quad adds
// its four sums up (three four-lane additions) and stores ONE PRIMARY sum per bucket; the entry count in the task's meta carries bit 31, which makes the
// two reduction kernels treat every bucket as whole.  Hot buckets (repeated scalars) fail the test and keep the equal pieces.
template <uint32_t FB, int MINW = 2, bool OWNED = false>
__global__ void __launch_bounds__(MSM_THREADS, MINW) msm_accumulate(const G1Affine *bases, const uint32_t *entries,
                                                                  const uint32_t *bin_start, const uint32_t *task_start,
                                                                  XyzzW *partials, uint32_t *task_meta, MsmParams p) {
    constexpr uint32_t FINE = Shape<FB>::FINE, SLOT_PRIMARY = Shape<FB>::SLOT_PRIMARY, SLOT_HEAD = Shape<FB>::SLOT_HEAD,
                       SLOT_TAIL = Shape<FB>::SLOT_TAIL, SLOTS_PER_TASK = Shape<FB>::SLOTS_PER_TASK, META_PER_TASK = Shape<FB>::META_PER_TASK;
    extern __shared__ uint32_t sorted[];                      // [CHUNK]
    __shared__ uint32_t cnt[FINE], start[FINE + 1], cursor[FINE], max_cnt;
    const uint32_t tid = threadIdx.x, task = blockIdx.x;
    const uint32_t total_bins = p.batch * p.groups * p.nbins;
    if (task >= task_start[total_bins]) return;
    uint32_t blo = 0, bhi = total_bins;                       // bin = last index with task_start[bin] <= task
    // (one task per bin is the common case — uniform scalars at 2^20 —: then bin == task, two independent loads instead of
    //  the ten dependent ones of the search, which were ~10 us of a ~580 us task)
    if (task < total_bins && task_start[task] <= task && task_start[task + 1] > task) { blo = task; bhi = task + 1; }
    while (bhi - blo > 1) { uint32_t mid = (blo + bhi) >> 1; if (task_start[mid] <= task) blo = mid; else bhi = mid; }
    const uint32_t bin = blo, slice = task - task_start[bin];
    const uint32_t bs = bin_start[bin], be = bin_start[bin + 1];
    // a bin that needs k tasks is cut into k EQUAL slices (not CHUNK, CHUNK, ..., remainder): with ~6 tasks per
    // workgroup slot a mix of full and quarter-size tasks left the last full ones running alone (measured at 2^21:
    // 3.7 ms instead of 2.6 ms for the same additions)
    const uint32_t k_bin = task_start[bin + 1] - task_start[bin], per = (be - bs + k_bin - 1) / k_bin;
    const uint32_t s = bs + slice * per < be ? bs + slice * per : be, e = (s + per < be) ? s + per : be, nc = e - s;

    if (tid < FINE) { cnt[tid] = 0; cursor[tid] = 0; }
    __syncthreads();
    // Both passes of the sort work from registers: a lane's <= 64 entries are loaded up front with all loads in flight (one
    // dependent global load per loop iteration left the wave waiting on memory 64 times per pass, and at the start of a launch
    // every workgroup of the chip is in this phase); the registers are free here, the accumulator state is not live yet.
    constexpr uint32_t PER_LANE = CHUNK / MSM_THREADS;                    // 64
    uint32_t ent[PER_LANE];
#pragma unroll
    for (uint32_t k = 0; k < PER_LANE; k++) { const uint32_t idx = tid + k * MSM_THREADS; ent[k] = idx < nc ? entries[s + idx] : 0u; }
    // A slice of ONE hot bucket (repeated scalars: all-(r-1), or a witness that is one value) makes all 64 lanes of every LDS atomic below
    // hit the same word — a 64-way serialised read-modify-write, twice per entry: +10 % on the whole kernel, measured
    // (profiles/r05_msm_all_r_minus_1_kernels.txt).  A wave whose entries all carry the same fine bucket counts them in one atomic
    // and numbers them by position instead; the test is ~2 cheap instructions per entry in registers and one ballot.
    const uint32_t wave_first = (tid & ~63u);                              // index of this wave's first entry of round k = 0
    uint32_t f0, diff = 0;
    {
        const uint32_t e0 = __builtin_amdgcn_readfirstlane(ent[0]);       // (lane 0 of the wave: valid whenever the wave has any entry)
        f0 = e0 & (FINE - 1);
#pragma unroll
        for (uint32_t k = 0; k < PER_LANE; k++) if (tid + k * MSM_THREADS < nc) diff |= (ent[k] ^ e0) & (FINE - 1);
    }
    const bool wave_hot = wave_first < nc && __ballot(diff != 0) == 0;     // wave-uniform
    // valid entries of this wave: (k, lane) with wave_first + lane + k * MSM_THREADS < nc — a prefix in (k, lane) order
    uint32_t wave_n = 0;
    if (wave_hot) {
#pragma unroll
        for (uint32_t k = 0; k < PER_LANE; k++) {
            const uint32_t b0 = wave_first + k * MSM_THREADS;
            wave_n += b0 >= nc ? 0u : (nc - b0 < 64u ? nc - b0 : 64u);
        }
        if ((tid & 63) == 0) atomicAdd(&cnt[f0], wave_n);
    } else {
#pragma unroll
        for (uint32_t k = 0; k < PER_LANE; k++) if (tid + k * MSM_THREADS < nc) atomicAdd(&cnt[ent[k] & (FINE - 1)], 1u);
    }
    __syncthreads();
    if (tid < 64) {                                           // exclusive scan of the FINE counts by one wave
        constexpr uint32_t PER = FINE / 64;                   // 1 or 2 buckets per lane
        uint32_t own[PER], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) { own[k] = cnt[PER * tid + k]; sum += own[k]; }
        uint32_t v = sum;
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(v, off); if ((int)tid >= off) v += t; }
        uint32_t run = v - sum;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) { start[PER * tid + k] = run; run += own[k]; }
        if (tid == 63) start[FINE] = v;
        if (OWNED) {
            uint32_t mx = own[0];
#pragma unroll
            for (uint32_t k = 1; k < PER; k++) mx = own[k] > mx ? own[k] : mx;
            for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_xor(mx, off); mx = t > mx ? t : mx; }
            if (tid == 0) max_cnt = mx;
        }
    }
    __syncthreads();
    uint32_t *meta = task_meta + (size_t)task * META_PER_TASK;
    const uint32_t mu = (nc + MSM_THREADS - 1) / MSM_THREADS;
    const bool owned = OWNED && FINE * 4 == MSM_THREADS && max_cnt <= 8 * mu + 24;     // (workgroup-uniform)
    if (tid <= FINE) meta[tid] = start[tid];
    if (tid == 0) meta[FINE + 1] = nc | (owned ? TASK_OWNED_BIT : 0u);
    if (wave_hot) {
        uint32_t base = 0;
        if ((tid & 63) == 0) base = atomicAdd(&cursor[f0], wave_n);
        base = start[f0] + __builtin_amdgcn_readfirstlane(base);
#pragma unroll
        for (uint32_t k = 0; k < PER_LANE; k++)
            if (tid + k * MSM_THREADS < nc) sorted[base + k * 64 + (tid & 63)] = ent[k];          // (valid entries are a prefix in (k, lane) order)
    } else {
#pragma unroll
        for (uint32_t k = 0; k < PER_LANE; k++)
            if (tid + k * MSM_THREADS < nc) { const uint32_t f = ent[k] & (FINE - 1); sorted[start[f] + atomicAdd(&cursor[f], 1u)] = ent[k]; }
    }
    __syncthreads();
    if (nc == 0 || p.debug == 3) return;                      // (debug 3: time the sort alone)
    XyzzW *out = partials + (size_t)task * SLOTS_PER_TASK;
    const uint32_t imask = (1u << p.nbits) - 1;
    auto point_of = [&](uint32_t entry) -> const G1Affine * {          // (copy j, index i) -> address in the table
        const uint32_t t = entry >> 8;
        return bases + (size_t)(t >> p.nbits) * p.copy_stride + (t & imask);
    };
    // one flat loop over the lane's piece: every lane of the wave executes the same number of mixed additions
    // in lockstep; a bucket boundary only costs the (divergent) 144-byte flush of the finished accumulator.
    // (The same loop — one mixed-addition site — serves the owned shape: first / step / last differ, and no bucket boundary is ever met.)
    const bool own = OWNED && owned;
    const uint32_t step = own ? 4u : 1u;
    uint32_t lo, hi;
    if (own) { lo = start[tid >> 2] + (tid & 3u); hi = start[(tid >> 2) + 1]; }
    else {
        lo = tid * mu < nc ? tid * mu : nc; hi = lo + mu < nc ? lo + mu : nc;
        if (lo >= hi) return;
    }
    uint32_t en = 0, b = tid >> 2, bend = 0xffffffffu, run_start = lo;
    G1Affine pt;
    if (lo < hi) { en = sorted[lo]; pt = load_affine(point_of(en)); }
    if (!own) { b = en & (FINE - 1); bend = start[b + 1]; }
    XyzzW acc = xyzzw_identity();
    for (uint32_t i = lo; i < hi; i += step) {
        const uint32_t e_cur = en;
        AffW cur; cur.x = unpack<FqW>(pt.x); cur.y = unpack<FqW>(pt.y);
        if (i + step < hi) { en = sorted[i + step]; pt = load_affine(point_of(en)); }   // prefetch the next gather
        if (i == bend) {                                      // the previous bucket ended inside this piece
            const bool from_prev = (run_start == lo) && (start[b] < lo);
            store_xyzzw(out + (from_prev ? SLOT_HEAD + tid : SLOT_PRIMARY + b), acc);
            acc = xyzzw_identity();
            run_start = i; b = e_cur & (FINE - 1); bend = start[b + 1];
        }
        if (p.debug != 1) xyzzw_add_mixed(acc, cur, (e_cur & 0x80u) != 0);
    }
    if (OWNED && own) {                                       // the quad's four sums -> one PRIMARY sum per bucket
        const uint32_t role = tid & 3u;                       // (distributed form, ec29_quad_dev.h: lane r of the quad holds coordinate r)
        const FqW9 c1 = quad_distribute<1>(acc, role), c2 = quad_distribute<2>(acc, role), c3 = quad_distribute<3>(acc, role);
        FqW9 X = quad_distribute<0>(acc, role);
        for (int k = 1; k < 4; k++) X = xyzzw_add_dist(X, wsel(k == 1, c1, wsel(k == 2, c2, c3)), role);     // (one addition site)
        if (start[b + 1] > start[b]) store_coord(out + SLOT_PRIMARY + b, role, X);
        return;
    }
    const bool from_prev = (run_start == lo) && (start[b] < lo), into_next = bend > hi;
    store_xyzzw(out + (from_prev ? SLOT_HEAD + tid : (into_next ? SLOT_TAIL + tid : SLOT_PRIMARY + b)), acc);
}


int32_t msm_accumulate_prepare() {
    static std::atomic<bool> attr_set{false};                 // (several contexts may commit from several host threads)
    if (attr_set) return PLK_OK;
    PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_accumulate<6>), hipFuncAttributeMaxDynamicSharedMemorySize, CHUNK * sizeof(uint32_t)));
    PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_accumulate<7>), hipFuncAttributeMaxDynamicSharedMemorySize, CHUNK * sizeof(uint32_t)));
    PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_accumulate<6, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, CHUNK * sizeof(uint32_t)));
    PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_accumulate<6, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CHUNK * sizeof(uint32_t)));
    attr_set = true;
    return PLK_OK;
}

void msm_accumulate_launch(uint32_t fine_bits, int variant, uint32_t max_tasks, hipStream_t stream, const G1Affine *bases, const uint32_t *entries,
                           const uint32_t *bin_start, const uint32_t *task_start, XyzzW *partials, uint32_t *task_meta, const MsmParams &p) {
    const size_t lds = CHUNK * sizeof(uint32_t);
    if (fine_bits == 6 && variant == 2)
        hipLaunchKernelGGL((msm_accumulate<6, 2, true>), dim3(max_tasks), dim3(MSM_THREADS), lds, stream, bases, entries, bin_start, task_start, partials, task_meta, p);
    else if (fine_bits == 6 && variant == 1)
        hipLaunchKernelGGL((msm_accumulate<6, 1>), dim3(max_tasks), dim3(MSM_THREADS), lds, stream, bases, entries, bin_start, task_start, partials, task_meta, p);
    else if (fine_bits == 6)
        hipLaunchKernelGGL((msm_accumulate<6>), dim3(max_tasks), dim3(MSM_THREADS), lds, stream, bases, entries, bin_start, task_start, partials, task_meta, p);
    else
        hipLaunchKernelGGL((msm_accumulate<7>), dim3(max_tasks), dim3(MSM_THREADS), lds, stream, bases, entries, bin_start, task_start, partials, task_meta, p);
}

}  // namespace plk
