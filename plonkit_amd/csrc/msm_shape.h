// Declarations shared by msm.hip (recoding, partition, bucket reduction, drivers) and msm_accumulate.hip (kernel A, the
// dominant kernel of the library, a translation unit of its own since round 4 so that it can be compiled with the scheduling
// strategy that suits it — see msm_accumulate.hip).
#pragma once
#include "ctx.h"
#include "ec_dev.h"
#include "ec29_dev.h"

namespace plk {

constexpr int MSM_THREADS = 256;
// Buckets per accumulate workgroup = 2^FB ("fine" part of the bucket index; the coarse part selects the bin).  Two shapes
// are compiled (template parameter FB of the kernels below) and chosen per commitment by pick_fine_bits():
//   FB = 6: 64 buckets per task.  At c = 17 that is 1024 coarse bins of ~15 K entries at 2^20 terms = ONE task per bin,
//           so every bucket is reduced once (65 K task-buckets) — the bucket reduction (msm_task_reduce) is not a latency
//           detail: in VALU work it was 38 % of the accumulation (992 waves x 37 dependent full additions of 14 products
//           against 15.7 M mixed additions of 9.3), and it shares the GPU with the next commitment's accumulation.
//   FB = 7: 128 buckets per task, 512 bins of ~31 K entries = two tasks per bin (131 K task-buckets): the round-1 shape.
constexpr uint32_t FINE_BITS_MAX = 7;
constexpr uint32_t CHUNK = 16384;                // entries per accumulate workgroup (sorted in 64 KB of LDS; two workgroups per CU)
constexpr uint32_t DIGIT_CHUNK = 16384;            // scalars per partition workgroup (per window): 64 KB of LDS staging
constexpr uint32_t TASK_MAX = CHUNK;
                   // per-chunk bucket population handled cooperatively

constexpr uint32_t MSM_MAX_BATCH = 8;              // commitments sharing one pass over the same bases

struct MsmParams {
    uint32_t n;
    uint32_t c;               // window bits
    uint32_t windows;         // W per commitment
    uint32_t fine_bits;       // FB: buckets per accumulate task = 2^FB
    uint32_t coarse_bits;     // c - 1 - fine_bits
    uint32_t nbins;           // 1 << coarse_bits
    uint32_t batch;           // number of scalar vectors (same n, same bases); "global window" = m * W + w
    uint32_t debug;           // experiments only: 1 = skip the additions (sort cost), 0 = normal
    // Shifted copies of the bases (fixed-base precomputation): window w = j*groups + g takes its points from
    // copy j*groups of the table (2^(16*j*groups) * P_i) and drops them into bucket set g, so only `groups`
    // bucket sets have to be reduced and only c*groups doublings are left for the host Horner.
    uint32_t groups;          // bucket sets per commitment (= windows when there is one copy)
    uint32_t nbits;           // entry index = (j << nbits) | i
    uint32_t copy_stride;     // points between table copies j and j+1 (= groups * srs_n)
};

struct ScalarSet { const Fr *v[MSM_MAX_BATCH]; };

// ---- scalar recoding shared by msm.hip (fused pre-phase) and msm_small.hip
__device__ __forceinline__ uint32_t extract_bits(const uint32_t *k, uint32_t pos, uint32_t c) {
    uint32_t limb = pos >> 5, off = pos & 31;
    if (limb >= 8) return 0;
    uint64_t v = k[limb];
    if (limb + 1 < 8) v |= (uint64_t)k[limb + 1] << 32;
    return (uint32_t)(v >> off) & ((1u << c) - 1);
}
constexpr int RC_WINDOWS = 15;                      // 17-bit windows over the 254-bit scalars
// the signed digits of msm_digits for c = 17, 15 windows, in registers: d[w] in [-2^16, 2^16], top window unsigned
__device__ __forceinline__ void recode17(const Fr &k, int32_t (&d)[RC_WINDOWS]) {
    uint32_t carry = 0;
#pragma unroll
    for (uint32_t w = 0; w < RC_WINDOWS; w++) {
        const uint32_t v = extract_bits(k.l, w * 17, 17) + carry;
        if (v >= (1u << 16) && w + 1 < RC_WINDOWS) { d[w] = (int32_t)v - (1 << 17); carry = 1; } else { d[w] = (int32_t)v; carry = 0; }
    }
}

// Per-task output of kernel A: 128 PRIMARY slots (a bucket whose run lies inside one lane's slice), and per
// lane one HEAD slot (its first run continues a bucket begun by an earlier lane) and one TAIL slot (its last
// run is continued by a later lane).  Which slots are live follows from the bucket offsets alone.
template <uint32_t FB> struct Shape {
    static constexpr uint32_t FINE = 1u << FB;
    static constexpr uint32_t SLOT_PRIMARY = 0, SLOT_HEAD = FINE, SLOT_TAIL = FINE + MSM_THREADS, SLOTS_PER_TASK = FINE + 2 * MSM_THREADS;
    static constexpr uint32_t META_PER_TASK = FINE + 2;   // start[0..FINE] and the entry count
};

// bit 31 of a task's entry count in its meta block: the task's buckets were accumulated by the lanes that own them (msm_accumulate<.., OWNED>): one PRIMARY
// sum per bucket, no HEAD / TAIL pieces
constexpr uint32_t TASK_OWNED_BIT = 0x80000000u;

// kernel A (msm_accumulate.hip): one workgroup per task; fine_bits 6 or 7; variant 1 = the 512-register / one-wave-per-SIMD build
// of the fine_bits-6 shape (measurement knob PLK_MSM_ONE_WAVE), variant 2 = the build that lets a task's lanes own its buckets when they are
// evenly filled (commitments of <= 2^16 terms), 0 = the plain kernel
int32_t msm_accumulate_prepare();                                   // dynamic-LDS attributes, once per process
void msm_accumulate_launch(uint32_t fine_bits, int variant, uint32_t max_tasks, hipStream_t stream, const G1Affine *bases, const uint32_t *entries,
                           const uint32_t *bin_start, const uint32_t *task_start, XyzzW *partials, uint32_t *task_meta, const MsmParams &p);

}  // namespace plk
