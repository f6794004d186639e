// Execution context: one GPU, one stream, cached twiddle tables, resident SRS, scratch arenas.
// Stands where bellman_ce's `Worker` stands in the reference (src/plonk.rs:41,47,183).
#pragma once
#include <utility>
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include "../../include/plonkit_amd.h"
#include "field_dev.h"
#include "hostmath.h"

namespace plk {

void set_error(const std::string &msg);
int32_t hip_fail(hipError_t e, const char *what, const char *file, int line);

#define PLK_HIP(expr)                                                          \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) return plk::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define PLK_TRY(expr)                         \
    do {                                      \
        int32_t _rc = (expr);                 \
        if (_rc != PLK_OK) return _rc;        \
    } while (0)

constexpr uint32_t MAX_LOG_N = 28;          // 2-adicity of Fr
constexpr uint32_t POW_SPLIT = 14;          // two-level power tables: e = hi * 2^14 + lo
constexpr uint32_t POW_TAB = 1u << POW_SPLIT;

// base^e for e < 2^28 as lo[e & 16383] * hi[e >> 14]
struct PowTable {
    const Fr *lo = nullptr;
    const Fr *hi = nullptr;
    const uint32_t *hi_sliced = nullptr;     // optional: the hi table as 9 x 29-bit limbs, 48 B per entry (NTT stage twiddles)
    const uint32_t *hi_tw3 = nullptr;        // optional: the hi table as three shifted copies per entry (field29_dev.h mul_tw3), 112 B per entry
};

// grows-only device buffer.  `borrowed`: a view of another context's allocation (plk_ctx_share_srs) — never freed here,
// never grown (reserve() beyond the capacity fails).
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool borrowed = false;
    int32_t reserve(size_t bytes);
    void release();
    void borrow(const DevBuf &o) { release(); p = o.p; cap = o.cap; borrowed = o.p != nullptr; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace plk

struct plk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t flag_ready = nullptr;         // the satisfiability verdict of plk_prove has reached `pinned`
    int num_cus = 0;
    // NTT tables (device): omega_{2^28} powers forward / inverse, coset generator 7 and 7^-1
    plk::DevBuf tables;
    plk::PowTable tw_fwd, tw_inv;            // omega_{2^28}^{+-e}, external Montgomery form (R = 2^256)
    plk::PowTable tw_fwd_w, tw_inv_w;        // the same powers in the 2^261 domain of field29_dev.h (NTT passes)
    std::map<std::vector<uint32_t>, plk::PowTable> coset_tabs;   // keyed by the 8 limbs of the shift
    std::vector<void *> coset_allocs;
    std::map<uint32_t, void *> ntt_direct;                       // ntt.hip: inter-pass twiddle tables by (direction, digit plan)
    size_t ntt_direct_bytes = 0;                                 // what they hold together (capped: ntt.hip, PLK_NTT_DIRECT_CAP_MB)
    std::map<uint32_t, uint64_t> ntt_direct_used;                // per table: value of ntt_direct_clock at its last request (eviction order)
    std::map<uint32_t, size_t> ntt_direct_size;
    uint64_t ntt_direct_clock = 0;
    struct CosetDirect { plk::DevBuf buf; uint32_t log_n = 0; } coset_direct[2];   // ntt.hip: coset-shift powers of the extensions as tables ([1]: inverse)
    std::map<std::vector<uint32_t>, plk::Fr> inv_cache;
    plk::Fr n_inv[plk::MAX_LOG_N + 1];      // 2^-k
    plk::Fr n_inv_w[plk::MAX_LOG_N + 1];    // 2^-k in the 2^261 domain
    plk::DevBuf ntt_scratch[2];              // ping-pong buffers of the NTT passes: [0] main stream, [1] the prover's background stream
    hipStream_t bg_stream = nullptr;         // low-priority stream of the prover: extensions that no challenge waits for
    hipEvent_t bg_go = nullptr, bg_done = nullptr;
    // SRS
    const void *srs = nullptr;               // device, Montgomery affine, 64 B per point
    uint64_t srs_n = 0;
    plk::DevBuf srs_own;
    plk::DevBuf srs_w;                       // the same points in the 2^261 Montgomery domain of field29_dev.h (MSM gathers)
    bool srs_w_valid = false;
    uint32_t srs_w_copies = 0;               // shifted copies 2^(16k) * P held in srs_w (fixed-base table of the MSM)
    // optional second key: Crs<E, CrsForLagrangeForm> (L_i(tau)*G), used by plk_prove for commit_using_values
    // (`prove -l`, src/plonk.rs:138-146).  Same fields as above; SrsSlotSwap makes it the active key for one call.
    struct LagrangeKey {
        const void *pts = nullptr; uint64_t n = 0;
        plk::DevBuf own, w; bool w_valid = false; uint32_t w_copies = 0;
    } lag;
    // MSM: up to three commitments (or batches) may be in flight, each with its own scratch, result buffer and stream:
    // the latency-bound tail of commitment k-1 (bucket reduction) and the digit / partition kernels of k+1 then share the
    // GPU with the accumulation of k, which starts the moment the previous accumulation ends.  A slot is taken lowest
    // index first, so a caller that keeps at most two in flight (the prover) never allocates the third slot's scratch.
    static constexpr uint32_t MSM_SLOTS = 3;
    struct MsmSlot {
        plk::DevBuf a, b, c, d, e, f;
        void *pinned = nullptr; size_t pinned_cap = 0;
        uint32_t windows = 0, c_bits = 0, pending_parts = 0, batch = 1, roles = 2, fine_bits = 7;
        hipStream_t stream = nullptr;        // the kernels of this commitment; ordered after the caller's stream by `ready`
        hipEvent_t ready = nullptr;
        hipEvent_t acc_done = nullptr;       // recorded behind msm_accumulate: from here on the commitment only runs its latency-bound reduction
        hipEvent_t ev[2] = {nullptr, nullptr};   // optional bracket around msm_accumulate (bench roofline)
        bool busy = false;
        // short-commitment path (msm_small.hip): what msm_finish_batch needs to run the ordinary pipeline if a bucket list overflowed
        bool small = false;
        const void *fb_bases = nullptr, *fb_scalars[8] = {nullptr};
        uint64_t fb_srs_n = 0, fb_n = 0;
        uint32_t fb_copies = 0, fb_cbits = 0, fb_nbits = 0;
    } slot[MSM_SLOTS];
    uint64_t msm_enq = 0, msm_fin = 0;       // FIFO counters; commitment number k lives in slot[fifo[k % MSM_SLOTS]]
    uint8_t fifo[MSM_SLOTS] = {};
    uint32_t last_slot = 0;                  // slot of the commitment finished last (plk_msm_last_kernel_ms)
    MsmSlot &front_slot() { return slot[fifo[msm_fin % MSM_SLOTS]]; }
    plk::DevBuf prove_ws;                    // workspace of the prover rounds (grows only)
    plk::DevBuf poly_tmp, poly_tmp2;         // scan block totals / evaluation partials
    plk::DevBuf stage;                       // host<->device staging for the host-pointer API
    void *pinned = nullptr;                  // small pinned host buffer for results
    size_t pinned_cap = 0;
    void *pinned2 = nullptr;                 // pinned staging of the prover's temporaries
    size_t pinned2_cap = 0;
    std::vector<double> timings;
    // intermediate vectors of the last plk_prove, still resident in prove_ws (plk_prove_trace: test / debugging hook)
    struct Trace { const plk::Fr *ptr[12] = {nullptr}; uint64_t len[12] = {0}; bool valid = false; } trace;
    bool ev_on = false;                      // record the per-slot event bracket around msm_accumulate
    // multi-GPU commitments of the prover (plk_set_commit_shard): global index of the first resident SRS point and
    // the caller's all-ranks combiner for the Jacobian partial sums
    uint64_t shard_first = 0;
    plk_combine_fn combine = nullptr;
    void *combine_user = nullptr;
    uint32_t scatter_open = 0;               // owner-computes mode: vectors of the batch already sent to the workers whose all-gather has not run yet (prover.hip)
    void *comm = nullptr;                    // built-in communicator (plk_comm_init / _tcp, comm.cpp), or null
    std::vector<plk::host::HJac> commit_pieces;   // partial sums of a commitment longer than one MSM call (prover.hip)
    std::vector<plk::host::HJac> commit_done;     // finished commitments of a batch that is processed one at a time
    // plk_ctx_share_srs: a context that proves beside another one on the same GPU borrows that one's resident key(s) and MSM
    // fixed-base table(s) instead of holding a second copy.  The lender counts its borrowers and refuses to replace a key
    // while one exists; a borrower drops the loan when it is given a key of its own or destroyed.
    plk_ctx *srs_lender = nullptr;
    std::atomic<int> srs_borrowers{0};
    bool lag_borrowed = false;               // borrower: the Lagrange-form key in `lag` is the lender's (recorded when the loan was made:
                                             // the lender swaps its own key slots during a proof, so comparing pointers later is a race)
    std::atomic<int> lag_borrowers{0};       // lender: how many borrowers hold its Lagrange-form key (it may install one while none does)
    bool zombie = false;                     // lender destroyed while borrowers exist: only the lent key and tables are left, freed with the last loan
};

namespace plk {
// every entry point that replaces a resident key goes through these two: the MSM table of the old key is void, and a table
// (or key) that was only borrowed must not be written to when the next one is built
int32_t srs_replace_guard(plk_ctx *c, const char *who, bool lagrange_only = false);   // runtime.hip: PLK_ERR_ARG while another context borrows this one's key
inline void srs_table_invalidate(plk_ctx *c) { c->srs_w_valid = false; if (c->srs_w.borrowed) c->srs_w.release(); }
inline void lag_table_invalidate(plk_ctx *c) { c->lag.w_valid = false; if (c->lag.w.borrowed) c->lag.w.release(); }
void srs_return_loan(plk_ctx *c);                                 // runtime.hip
void srs_make_loan(plk_ctx *dst, plk_ctx *src);                   // runtime.hip
bool srs_orphan_lender(plk_ctx *c);                               // runtime.hip

// makes the Lagrange key the context's active SRS for the lifetime of the guard (host-side pointer swap only;
// kernels already enqueued keep the addresses they were launched with)
struct SrsSlotSwap {
    plk_ctx *c; bool on;
    SrsSlotSwap(plk_ctx *ctx, bool lagrange) : c(ctx), on(lagrange) { flip(); }
    ~SrsSlotSwap() { flip(); }
    void flip() {
        if (!on) return;
        std::swap(c->srs, c->lag.pts); std::swap(c->srs_n, c->lag.n);
        std::swap(c->srs_own, c->lag.own); std::swap(c->srs_w, c->lag.w);
        std::swap(c->srs_w_valid, c->lag.w_valid); std::swap(c->srs_w_copies, c->lag.w_copies);
    }
};
}  // namespace plk
