// Setup polynomials, verification key and the five prover rounds — device resident.
//
// Mirrors (reference file:line):
//   SetupForProver::prepare_setup_for_prover  src/plonk.rs:97-119   -> plk_setup_prepare
//   make_verification_key + vk.write          src/plonk.rs:122-124 ; src/bin/main.rs:501-502 -> plk_setup_write_vk
//   SetupForProver::prove("keccak"), monomial key = prove_by_steps  src/plonk.rs:132-159 -> plk_prove
//   Proof::write                              src/bin/main.rs:407-408 (layout SURVEY.md A.1)
// The protocol (bellman_ce better_cs prover, source absent) is the one recovered in SURVEY.md
// Appendix A.3/A.4 and pinned byte-for-byte by test/circuits/simple/{vk,proof}.bin.
// Everything O(N) runs on the GPU (ntt.hip, msm.hip, poly.hip); the host keeps the transcript,
// a few dozen scalars per round and the circuit synthesis.
// Streams of a proof: ctx->stream carries the round-critical kernels; every commitment runs on its MSM slot's stream
// (msm.hip); ctx->bg_stream (own NTT scratch) carries the work no challenge waits for — the coset-major extensions of the
// wires, z and the public inputs, and the second opening quotient — started by hand behind the accumulation of the commitment
// in flight, so that it fills the latency-bound ends of a commitment instead of standing between two rounds.
#include "ctx.h"
#include "ntt.h"
#include "msm.h"
#include "poly.h"
#include "circuit.h"
#include "keccak.h"
#include "comm.h"
#include <chrono>
#include <memory>
#include <thread>
#include <mutex>
#include <algorithm>
#include <cstring>

namespace plk {
using namespace host;

static const uint64_t NON_RESIDUES[4] = {1, 5, 7, 10};     // k_j (vk.bin stores 5, 7, 10); k_quotient (poly.hip) forms 5x, 7x, 10x by shifts

static inline Fr to_dev(const HFr &h) { Fr f; memcpy(f.l, h.l, 32); return f; }
int32_t ensure_pinned2(plk_ctx *ctx, size_t bytes);
static inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static HFr host_omega(uint32_t log_n) {
    Fr w = ntt_omega(log_n);
    HFr h; memcpy(h.l, w.l, 32); return h;
}

// KZG commitments of vectors resident on the device: up to 8 over the same SRS prefix share one pass of the MSM
// kernels.  lagrange = commit_using_values: the vectors are evaluations over the domain and the bases the resident
// Lagrange-form key (same group element).
//
// Split form: the commitment's kernels run on their own stream (msm.hip, MsmSlot), so independent work issued on
// ctx->stream between begin and end fills the SIMDs that the bucket-reduction phase leaves idle.
//
// Multi-GPU (plk_set_commit_shard): this rank holds the SRS points [first, first + srs_n) only, commits that index
// range of every vector and hands the Jacobian partial sums to the caller's combiner (all_gather + EC sum over RCCL).
// One MSM call takes at most 2^24 terms (24-bit index field of a bucket entry): longer vectors (domains 2^25, 2^26 —
// SETUP_MAX_POW2 of the reference is 26) are committed in pieces against successive SRS ranges, summed on the host.
static uint64_t msm_piece_terms() {
    static const uint64_t v = [] { const char *e = getenv("PLK_MSM_MAX_TERMS"); uint64_t x = e ? strtoull(e, nullptr, 10) : 0; return x >= 4096 && x <= (1ull << 24) ? x : (1ull << 24); }();
    return v;                                                 // (the environment override exists for the tests)
}
// enqueues the pieces of `cnt` commitments over scalars[lo + ...]; every piece but the last is finished here and summed
// into ctx->commit_pieces, the last one stays in flight (the caller overlaps it and calls commit_end)
static int32_t enqueue_pieces(plk_ctx *ctx, const Fr *const *vecs, uint32_t cnt, uint64_t lo, uint64_t hi, uint64_t first_base) {
    const uint64_t piece = msm_piece_terms();
    ctx->commit_pieces.clear();
    const Fr *shifted[8];
    for (uint64_t off = 0;; off += piece) {
        const uint64_t len = hi - lo - off < piece ? hi - lo - off : piece;
        for (uint32_t k = 0; k < cnt; k++) shifted[k] = vecs[k] + lo + off;
        PLK_TRY(msm_enqueue_batch(ctx, shifted, cnt, len, first_base + off, ctx->stream));
        if (off + len >= hi - lo) break;
        HJac j[8];
        PLK_TRY(msm_finish_batch(ctx, nullptr, j));
        if (ctx->commit_pieces.empty()) ctx->commit_pieces.assign(j, j + cnt);
        else for (uint32_t k = 0; k < cnt; k++) ctx->commit_pieces[k] = jac_add(ctx->commit_pieces[k], j[k]);
    }
    return PLK_OK;
}
static int32_t finish_pieces(plk_ctx *ctx, uint32_t cnt, HJac *j) {
    if (ctx->msm_fin != ctx->msm_enq && ctx->front_slot().batch != cnt) {      // never pop somebody else's commitment
        set_error("commitment FIFO out of step: the slot in flight holds a different batch than the prover enqueued"); return PLK_ERR_ARG; }
    PLK_TRY(msm_finish_batch(ctx, nullptr, j));
    if (!ctx->commit_pieces.empty()) {
        for (uint32_t k = 0; k < cnt; k++) j[k] = jac_add(j[k], ctx->commit_pieces[k]);
        ctx->commit_pieces.clear();
    }
    return PLK_OK;
}
static int32_t commit_begin(plk_ctx *ctx, const Fr *const *vecs, uint32_t count, uint64_t n, bool lagrange = false) {
    SrsSlotSwap active(ctx, lagrange);
    uint64_t lo = 0, hi = n;
    if (ctx->combine) {
        lo = ctx->shard_first < n ? ctx->shard_first : n;
        hi = ctx->shard_first + ctx->srs_n < n ? ctx->shard_first + ctx->srs_n : n;
        if (hi < lo) hi = lo;
    }
    const uint64_t first_base = ctx->combine ? 0 : lo;
    ctx->commit_done.clear();
    // owner-computes mode (comm.h): the other ranks do not run this prover — they get their slices of the vectors now, commit them
    // while this rank commits its own, and meet it in commit_end's exchange
    if (comm_scatter_owner(ctx)) {
        if (ctx->shard_first != 0) { set_error("scatter mode: the owner (rank 0) must hold the first slice of the key"); return PLK_ERR_ARG; }
        PLK_TRY(comm_send_work(ctx, reinterpret_cast<const void *const *>(vecs), count, n, ctx->srs_n, lagrange, ctx->stream));
        ctx->scatter_open = count;                                // the workers now sit in this batch's all-gather: it MUST be run (commit_end, or the FIFO guard)
    }
    // at the largest sizes a batch of commitments would need gigabytes of per-task partial-sum slots (2^26 gates: 16 GiB for
    // four wires, which is what stands between that domain and the 288 GB): one commitment at a time there
    if (count > 1 && hi - lo > msm_piece_terms()) {
        for (uint32_t k = 0; k + 1 < count; k++) {
            HJac j;
            PLK_TRY(enqueue_pieces(ctx, vecs + k, 1, lo, hi, first_base));
            PLK_TRY(finish_pieces(ctx, 1, &j));
            ctx->commit_done.push_back(j);
        }
        return enqueue_pieces(ctx, vecs + count - 1, 1, lo, hi, first_base);
    }
    return enqueue_pieces(ctx, vecs, count, lo, hi, first_base);
}
static int32_t commit_end(plk_ctx *ctx, uint32_t count, HAffine *out) {
    HJac j[8];
    if (!ctx->commit_done.empty()) {                              // the one-at-a-time path of commit_begin
        const uint32_t done = (uint32_t)ctx->commit_done.size();
        for (uint32_t k = 0; k < done; k++) j[k] = ctx->commit_done[k];
        PLK_TRY(finish_pieces(ctx, 1, j + done));
        ctx->commit_done.clear();
    } else PLK_TRY(finish_pieces(ctx, count, j));
    if (ctx->combine) {
        plk_g1_jacobian raw[8];
        for (uint32_t k = 0; k < count; k++) { memcpy(raw[k].x, j[k].x.l, 32); memcpy(raw[k].y, j[k].y.l, 32); memcpy(raw[k].z, j[k].z.l, 32); }
        set_error("");
        ctx->scatter_open = 0;
        const int32_t rc = ctx->combine(ctx->combine_user, raw, count);
        if (rc != PLK_OK) {                                        // (the built-in combiner says why: keep its words)
            const std::string why = plk_last_error();
            set_error(why.empty() ? std::string("commitment combiner (plk_set_commit_shard) failed") : "commitment combiner failed: " + why);
            return rc;
        }
        for (uint32_t k = 0; k < count; k++) { memcpy(j[k].x.l, raw[k].x, 32); memcpy(j[k].y.l, raw[k].y, 32); memcpy(j[k].z.l, raw[k].z, 32); }
    }
    jac_to_affine_batch(j, count, out);                       // one field inversion for the whole batch (13 us each on the host)
    return PLK_OK;
}
static int32_t commit_many(plk_ctx *ctx, const Fr *const *coefs, uint32_t count, uint64_t n, HAffine *out, bool lagrange = false) {
    for (uint32_t done = 0; done < count;) {
        uint32_t b = count - done > 8 ? 8 : count - done;
        PLK_TRY(commit_begin(ctx, coefs + done, b, n, lagrange));
        PLK_TRY(commit_end(ctx, b, out + done));
        done += b;
    }
    return PLK_OK;
}

// plk_comm_serve, the worker side of owner-computes mode: whatever the owner's commit_begin sends is committed against this context's
// slice of the key (commit_begin / commit_end on the received slices: same pieces, same FIFO, same exchange as a replicated rank)
static int32_t serve_impl(plk_ctx *ctx, uint64_t *batches) {
    for (;;) {
        ShardWork w;
        PLK_TRY(comm_recv_work(ctx, &w, ctx->stream));
        if (w.op == SHARD_STOP) return PLK_OK;
        const bool lagrange = w.lagrange != 0;
        if (lagrange && !ctx->lag.pts) { set_error("plk_comm_serve: the owner commits against a Lagrange-form key this rank does not hold"); return PLK_ERR_SRS; }
        {
            SrsSlotSwap active(ctx, lagrange);
            if (ctx->srs_n != w.slice) { set_error("plk_comm_serve: this rank's key slice has a different size than the owner's (every rank holds points [r L, (r+1) L))"); return PLK_ERR_SRS; }
        }
        const Fr *vecs[8];
        static const Fr none{};
        for (uint32_t k = 0; k < w.count; k++) vecs[k] = w.len ? static_cast<const Fr *>(w.vec[k]) : &none;
        // (the received slices ARE this rank's index range: commit them as vectors of their own length from index 0)
        const uint64_t keep_first = ctx->shard_first;
        ctx->shard_first = 0;
        int32_t rc = commit_begin(ctx, vecs, w.count, w.len, lagrange);
        HAffine out[8];
        if (rc == PLK_OK) rc = commit_end(ctx, w.count, out);
        ctx->shard_first = keep_first;
        PLK_TRY(rc);
        if (batches) ++*batches;
    }
}

// plk_prove / plk_setup_write_vk own the two-slot commitment FIFO for the duration of the call: it must be empty on
// entry (a commitment the caller enqueued with plk_msm_g1_enqueue_dev and never finished would otherwise be popped as
// one of the prover's), and whatever an early error return leaves in flight is drained before the call returns.
static int32_t fifo_must_be_empty(plk_ctx *ctx, const char *who) {
    if (ctx->msm_enq == ctx->msm_fin) return PLK_OK;
    set_error(std::string(who) + ": a commitment enqueued with plk_msm_g1_enqueue_dev is still in flight (call plk_msm_g1_finish first)");
    return PLK_ERR_ARG;
}
struct FifoGuard {
    plk_ctx *ctx;
    explicit FifoGuard(plk_ctx *c) : ctx(c) {}
    ~FifoGuard() {
        if (ctx->msm_fin != ctx->msm_enq) {
            const std::string keep = plk_last_error();               // the drain must not replace the error being returned
            while (ctx->msm_fin != ctx->msm_enq) { HJac j[8]; (void)msm_finish_batch(ctx, nullptr, j); }
            set_error(keep);
        }
        ctx->commit_done.clear(); ctx->commit_pieces.clear();
        // Owner-computes mode: a batch went out to the workers (commit_begin) and the call is returning without commit_end — "must satisfy",
        // a failed PLK_TRY in between.  The workers are committing their slices and will enter the batch's all-gather; if the owner never does,
        // they sit there for the exchange deadline (180 s) and the owner's next broadcast meets their all-gather as a mismatched collective.
        // Run the exchange with empty sums (nobody reads the result): owner and workers stay in step, the communicator stays usable.
        if (ctx->scatter_open && ctx->combine) {
            const std::string keep = plk_last_error();
            plk_g1_jacobian raw[8];
            memset(raw, 0, sizeof raw);
            (void)ctx->combine(ctx->combine_user, raw, ctx->scatter_open);
            set_error(keep);
        }
        ctx->scatter_open = 0;
    }
};

struct Arena {
    DevBuf *buf; size_t off = 0;
    template <class T> T *take(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        T *p = reinterpret_cast<T *>((char *)buf->p + off);
        off += bytes;
        return p;
    }
};

}  // namespace plk

struct plk_setup {
    uint64_t n = 0, N = 0, num_inputs = 0, n_real = 0, num_vars = 0, num_gates = 0;
    uint32_t log_n = 0;
    plk::DevBuf store;                     // one allocation holding everything below
    plk::Fr *sel_coef[7] = {nullptr}, *sel_vals[7] = {nullptr}, *sig_coef[4] = {nullptr}, *sig_vals[4] = {nullptr};
    uint32_t *gate_vars[4] = {nullptr};
    // coset LDEs (4N evaluations on 7*<omega_4N>) of the 7 selectors, 4 sigmas and L_0: circuit constants,
    // computed by the first proof and kept (the reference recomputes them in every prove_by_steps call
    // because plonkit passes `None` precomputations, src/plonk.rs:156; the values are identical)
    mutable plk::DevBuf lde_store;
    mutable plk::Fr *lde[13] = {nullptr};       // 7 selectors, 4 sigma, L0 (pre-scaled, see QuotientArgs), coset points
    mutable bool lde_ready = false;
    mutable bool zh_inv_ready = false;
    mutable plk::HFr zh_inv[4];
    mutable plk::HFr icoset_c[5];            // i^-1 (i = omega_4) and 7^(-N c) / 4, c = 0..3: constants of the coset iNTT's combine step
    mutable std::mutex lazy_mu;              // the cached extensions and constants above are filled by the FIRST proof: one setup may be
                                             // proved from several contexts / host threads at once (SetupForProver::prove takes &self)
    uint64_t num_circuit_vars = 0;         // circom wires; temporaries follow
    std::vector<plk::WitnessOp> ops;       // linear forms defining the transpiler's temporaries
    bool ops_independent = false;          // no temporary reads another temporary -> order-free evaluation
    bool ops_chained = false;              // a temporary reads at most its immediate predecessor (partial-sum chains of long linear
                                           // combinations): evaluated on the device run by run (run_start = first temporary of every run)
    std::vector<uint32_t> run_start;
    plk::DevBuf ops_dev, terms_dev, runs_dev;   // the records on the device (when ops_independent or ops_chained): evaluated there
    std::vector<plk::WitnessTerm> op_terms;
    plk::big_vector<plk::HFr> h_cols;      // host phase only: 7 selector columns x N, until plk_setup_upload
    plk::big_vector<uint32_t> h_vars;      // host phase only: 4 variable-index columns x N
};

using namespace plk;

void plk_circuit_unregister(plk_circuit *c) {
    if (c->witness_registered) { (void)hipHostUnregister((void *)c->witness.data()); c->witness_registered = false; }
}

extern "C" {

uint64_t plk_setup_domain_size(const plk_setup *s) { return s ? s->N : 0; }
void plk_setup_free(plk_setup *s) { if (s) { s->store.release(); s->lde_store.release(); s->ops_dev.release(); s->terms_dev.release(); s->runs_dev.release(); delete s; } }

// the domain size N (and log2 N) a circuit's setup will have — transpile only (pure CPU, ~18 ms at 2^20 gates), no columns, no device work: what
// `dump-lagrange` needs of prepare_setup_for_prover (src/bin/main.rs:360-381 builds the whole setup to read setup.n from it)
int32_t plk_circuit_domain_size(const plk_circuit *c, uint64_t *n_out) {
    if (!c || !n_out) { set_error("plk_circuit_domain_size: bad argument"); return PLK_ERR_ARG; }
    return guarded("plk_circuit_domain_size", PLK_ERR_ARG, [&]() -> int32_t {
        Transpiled T;
        T.collect_stats = false;
        if (!transpile(c->r1cs, nullptr, &T)) return PLK_ERR_UNSAT;
        const uint64_t n_real = (uint64_t)(c->r1cs.num_inputs - 1) + T.num_gates;
        uint64_t N = 1; uint32_t log_n = 0;
        while (N < n_real + 1) { N <<= 1; log_n++; }
        if (log_n + 2 > MAX_LOG_N) { set_error("setup power of two is not in the correct range"); return PLK_ERR_SIZE; }
        *n_out = N;
        return PLK_OK;
    });
}

// SetupForProver::prepare_setup_for_prover in two phases, so that a host program can run the first one (pure CPU:
// transpile, selector / variable-index columns) while the GPU side of its start-up is still under way on another thread
// (HIP initialisation, key upload, MSM table — the `plonkit` binary does exactly that), and the second one when both are done.
static int32_t setup_host_impl(const plk_circuit *c, plk_setup **out) {
    if (!c || !out) { set_error("plk_setup_prepare: bad argument"); return PLK_ERR_ARG; }
    *out = nullptr;
    const bool timing = getenv("PLK_CLI_TIMING") != nullptr;     // phase times of the setup on stderr (tools/cli_scale.sh)
    double t_last = now_ms();
    auto mark = [&](const char *what) { if (timing) { double t = now_ms(); fprintf(stderr, "[timing]   setup: %-22s +%.3f s\n", what, (t - t_last) / 1e3); t_last = t; } };
    Transpiled T;
    T.collect_stats = false;                                     // a million std::string names are only wanted by `analyse`
    if (!transpile(c->r1cs, nullptr, &T)) return PLK_ERR_UNSAT;
    mark("transpile");
    std::unique_ptr<plk_setup> S(new plk_setup());
    S->num_inputs = c->r1cs.num_inputs - 1;
    S->num_gates = T.num_gates;
    S->n_real = S->num_inputs + T.num_gates;
    S->num_vars = T.num_vars;
    uint64_t N = 1; uint32_t log_n = 0;
    while (N < S->n_real + 1) { N <<= 1; log_n++; }
    if (log_n + 2 > MAX_LOG_N) { set_error("setup power of two is not in the correct range"); return PLK_ERR_SIZE; }   // src/plonk.rs:109-112
    S->N = N; S->n = N - 1; S->log_n = log_n;
    S->num_circuit_vars = c->r1cs.num_variables;
    S->ops.swap(T.ops); S->op_terms.swap(T.op_terms);
    S->ops_independent = true;
    for (const WitnessTerm &t : S->op_terms) if (t.var >= c->r1cs.num_variables) { S->ops_independent = false; break; }
    if (!S->ops_independent) {
        // long linear combinations: temporary i (a partial sum) reads temporary i - 1 and circom wires only -> runs
        const uint64_t ncv0 = c->r1cs.num_variables;
        S->ops_chained = true;
        for (size_t i = 0; i < S->ops.size() && S->ops_chained; i++) {
            bool continues = false;
            for (uint32_t k = 0; k < S->ops[i].count; k++) {
                const uint32_t v = S->op_terms[S->ops[i].first + k].var;
                if (v < ncv0) continue;
                if (i > 0 && v == ncv0 + i - 1) continues = true;
                else { S->ops_chained = false; break; }
            }
            if (!continues) S->run_start.push_back((uint32_t)i);
        }
        if (!S->ops_chained) std::vector<uint32_t>().swap(S->run_start);
    }
    // rows of the trace: one gate per public input first (q_a = -1), then the transpiler's gates, then padding with the dummy
    // variable.  The gate pieces are read once, piece-parallel, straight into the seven selector columns and the four
    // variable-index columns.
    const size_t n_in = S->num_inputs, n_rows = n_in + T.num_gates;
    S->h_cols.resize((size_t)7 * N);
    S->h_vars.resize((size_t)4 * N);
    HFr *cols = S->h_cols.data(); uint32_t *vars = S->h_vars.data();
    const HFr minus_one = -HFr::one();
    for (size_t r = 0; r < n_in; r++) {
        for (int k = 0; k < 7; k++) cols[(size_t)k * N + r] = k == 0 ? minus_one : HFr::zero();
        vars[r] = (uint32_t)(r + 1); vars[N + r] = vars[2 * N + r] = vars[3 * N + r] = 0;
    }
    parallel_for(T.pieces.size(), 1, [&](size_t lo, size_t hi) {
        for (size_t p = lo; p < hi; p++) {
            const big_vector<Gate> &gs = T.pieces[p];
            const uint32_t shift = T.tmp_shift[p], first_tmp = (uint32_t)T.first_tmp;
            size_t r = n_in + T.gate0[p];
            for (size_t i = 0; i < gs.size(); i++, r++) {
                const Gate &g = gs[i];
                for (int k = 0; k < 7; k++) cols[(size_t)k * N + r] = g.q[k];
                for (int j = 0; j < 4; j++) vars[(size_t)j * N + r] = g.v[j] >= first_tmp ? g.v[j] + shift : g.v[j];
            }
        }
    });
    parallel_for(N - n_rows, 1 << 15, [&](size_t lo, size_t hi) {
        for (size_t r = n_rows + lo; r < n_rows + hi; r++) {
            for (int k = 0; k < 7; k++) cols[(size_t)k * N + r] = HFr::zero();
            for (int j = 0; j < 4; j++) vars[(size_t)j * N + r] = 0;
        }
    });
    mark("columns (host fill)");
    *out = S.release();
    return PLK_OK;
}

static int32_t setup_upload_impl(plk_ctx *ctx, plk_setup *S) {
    if (!ctx || !S) { set_error("plk_setup_upload: bad argument"); return PLK_ERR_ARG; }
    if (S->store.p) return PLK_OK;                               // already resident
    if (S->h_cols.empty()) { set_error("plk_setup_upload: the host phase has not run"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    const bool timing = getenv("PLK_CLI_TIMING") != nullptr;
    double t_last = now_ms();
    auto mark = [&](const char *what) { if (timing) { double t = now_ms(); fprintf(stderr, "[timing]   setup: %-22s +%.3f s\n", what, (t - t_last) / 1e3); t_last = t; } };
    const uint64_t N = S->N; const uint32_t log_n = S->log_n;
    static_assert(sizeof(WitnessOp) == 40 && sizeof(WitnessTerm) == 40, "records are uploaded as they are (poly.hip)");
    hipStream_t st = ctx->stream;
    auto fail = [&](int32_t code) { S->store.release(); S->ops_dev.release(); S->terms_dev.release(); S->runs_dev.release(); return code; };
    int32_t rc;
    if (S->ops_chained && !S->ops.empty()) {
        if ((rc = S->runs_dev.reserve(S->run_start.size() * sizeof(uint32_t))) != PLK_OK) return fail(rc);
        if (hipMemcpyAsync(S->runs_dev.p, S->run_start.data(), S->run_start.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st) != hipSuccess)
            return fail(hip_fail(hipGetLastError(), "H2D witness runs", __FILE__, __LINE__));
    }
    if ((S->ops_independent || S->ops_chained) && !S->ops.empty()) {                     // temporaries will be evaluated on the device
        if ((rc = S->ops_dev.reserve(S->ops.size() * sizeof(WitnessOp))) != PLK_OK) return fail(rc);
        if ((rc = S->terms_dev.reserve(S->op_terms.size() * sizeof(WitnessTerm) + 8)) != PLK_OK) return fail(rc);
        if (hipMemcpyAsync(S->ops_dev.p, S->ops.data(), S->ops.size() * sizeof(WitnessOp), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(S->terms_dev.p, S->op_terms.data(), S->op_terms.size() * sizeof(WitnessTerm), hipMemcpyHostToDevice, st) != hipSuccess)
            return fail(hip_fail(hipGetLastError(), "H2D witness ops", __FILE__, __LINE__));
    }
    Arena A{&S->store};
    size_t total = 22 * ((N * sizeof(Fr) + 255) & ~(size_t)255) + 4 * ((N * 4 + 255) & ~(size_t)255);
    if ((rc = S->store.reserve(total)) != PLK_OK) return fail(rc);
    for (int k = 0; k < 7; k++) S->sel_coef[k] = A.take<Fr>(N);
    for (int k = 0; k < 7; k++) S->sel_vals[k] = A.take<Fr>(N);
    for (int j = 0; j < 4; j++) S->sig_coef[j] = A.take<Fr>(N);
    for (int j = 0; j < 4; j++) S->sig_vals[j] = A.take<Fr>(N);
    for (int j = 0; j < 4; j++) S->gate_vars[j] = A.take<uint32_t>(N);
    // the columns are uploaded back to back and interpolated as they land
    for (int j = 0; j < 4; j++)
        if (hipMemcpyAsync(S->gate_vars[j], S->h_vars.data() + (size_t)j * N, N * 4, hipMemcpyHostToDevice, st) != hipSuccess) return fail(hip_fail(hipGetLastError(), "H2D vars", __FILE__, __LINE__));
    for (int k = 0; k < 7; k++) {
        if (hipMemcpyAsync(S->sel_vals[k], S->h_cols.data() + (size_t)k * N, N * sizeof(Fr), hipMemcpyHostToDevice, st) != hipSuccess) return fail(hip_fail(hipGetLastError(), "H2D selector", __FILE__, __LINE__));
        if (hipMemcpyAsync(S->sel_coef[k], S->sel_vals[k], N * sizeof(Fr), hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(hip_fail(hipGetLastError(), "D2D selector", __FILE__, __LINE__));
        if ((rc = ntt_dev(ctx, S->sel_coef[k], log_n, true, nullptr, st)) != PLK_OK) return fail(rc);
    }
    if (hipStreamSynchronize(st) != hipSuccess) return fail(hip_fail(hipGetLastError(), "sync", __FILE__, __LINE__));
    big_vector<HFr>().swap(S->h_cols);                           // the host columns go away here
    big_vector<uint32_t>().swap(S->h_vars);
    mark("selectors (upload + 7 iNTT)");
    // the permutation (rotate-left over each variable's occurrences) from the variable-index table, on the device (perm.hip)
    {
        Fr kk[4];
        for (int j = 0; j < 4; j++) kk[j] = from_u64<FrParams>(NON_RESIDUES[j]);
        DevBuf idx_buf;
        if ((rc = idx_buf.reserve((size_t)4 * N * 4)) != PLK_OK) return fail(rc);
        uint32_t *idx = idx_buf.as<uint32_t>();
        if ((rc = build_permutation_index(ctx, S->gate_vars, (uint32_t)N, S->num_vars, idx, st)) != PLK_OK) { idx_buf.release(); return fail(rc); }
        for (int j = 0; j < 4; j++) {
            if ((rc = sigma_from_index(S->sig_vals[j], idx + (size_t)j * N, (uint32_t)N, log_n, ctx->tw_fwd, kk, st)) != PLK_OK) { idx_buf.release(); return fail(rc); }
            if (hipMemcpyAsync(S->sig_coef[j], S->sig_vals[j], N * sizeof(Fr), hipMemcpyDeviceToDevice, st) != hipSuccess) { idx_buf.release(); return fail(hip_fail(hipGetLastError(), "D2D sigma", __FILE__, __LINE__)); }
            if (log_n > 22 && (rc = ntt_dev(ctx, S->sig_coef[j], log_n, true, nullptr, st)) != PLK_OK) { idx_buf.release(); return fail(rc); }
        }
        // (one launch per pass for the four of them where the batch scratch is small)
        if (log_n <= 22 && (rc = ntt_batch_dev(ctx, S->sig_coef, 4, log_n, true, nullptr, st, 0)) != PLK_OK) { idx_buf.release(); return fail(rc); }
        hipError_t e = hipStreamSynchronize(st);
        idx_buf.release();
        if (e != hipSuccess) return fail(hip_fail(e, "sync", __FILE__, __LINE__));
    }
    mark("permutation (device sort + 4 iNTT)");
    return PLK_OK;
}

static int32_t setup_prepare_impl(plk_ctx *ctx, const plk_circuit *c, plk_setup **out) {
    if (!ctx || !c || !out) { set_error("plk_setup_prepare: bad argument"); return PLK_ERR_ARG; }
    plk_setup *S = nullptr;
    PLK_TRY(setup_host_impl(c, &S));
    const int32_t rc = setup_upload_impl(ctx, S);
    if (rc != PLK_OK) { plk_setup_free(S); *out = nullptr; return rc; }
    *out = S;
    return PLK_OK;
}

static void put_u64(std::vector<uint8_t> &b, uint64_t v) { for (int i = 7; i >= 0; i--) b.push_back((uint8_t)(v >> (8 * i))); }
static void put_g1(std::vector<uint8_t> &b, const HAffine &p) { uint8_t t[64]; g1_to_bytes(p, t); b.insert(b.end(), t, t + 64); }
static void put_fr(std::vector<uint8_t> &b, const HFr &v) { uint8_t t[32]; v.to_be_bytes(t); b.insert(b.end(), t, t + 32); }

static int32_t setup_write_vk_impl(plk_ctx *ctx, const plk_setup *s, const uint8_t g2_bytes[256], uint8_t *out, uint64_t cap, uint64_t *len) {
    if (!ctx || !s || !g2_bytes || !out || !len) { set_error("plk_setup_write_vk: bad argument"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    if (!s->store.p) { set_error("plk_setup_write_vk: the setup is not on the device yet (plk_setup_upload)"); return PLK_ERR_ARG; }
    PLK_TRY(fifo_must_be_empty(ctx, "plk_setup_write_vk"));
    FifoGuard fifo_guard(ctx);
    std::vector<uint8_t> b;
    put_u64(b, s->n); put_u64(b, s->num_inputs);
    HAffine cm[11];
    {
        const Fr *polys[11];
        for (int k = 0; k < 7; k++) polys[k] = s->sel_coef[k];
        for (int j = 0; j < 4; j++) polys[7 + j] = s->sig_coef[j];
        PLK_TRY(commit_many(ctx, polys, 11, s->N, cm));
    }
    put_u64(b, 6);
    for (int k = 0; k < 6; k++) put_g1(b, cm[k]);
    put_u64(b, 1);
    put_g1(b, cm[6]);
    put_u64(b, 4);
    for (int j = 0; j < 4; j++) put_g1(b, cm[7 + j]);
    put_u64(b, 3);
    for (int j = 1; j < 4; j++) put_fr(b, HFr::from_u64(NON_RESIDUES[j]));
    b.insert(b.end(), g2_bytes, g2_bytes + 256);
    *len = b.size();
    if (b.size() > cap) { set_error("plk_setup_write_vk: buffer too small"); return PLK_ERR_ARG; }
    memcpy(out, b.data(), b.size());
    return PLK_OK;
}

int32_t plk_prove_timings(const plk_ctx *ctx, double *out_ms, uint32_t cap, uint32_t *count) {
    if (!ctx || !count) { set_error("plk_prove_timings: bad argument"); return PLK_ERR_ARG; }
    *count = (uint32_t)ctx->timings.size();
    for (uint32_t i = 0; i < *count && i < cap && out_ms; i++) out_ms[i] = ctx->timings[i];
    return PLK_OK;
}

static int32_t prove_impl(plk_ctx *ctx, const plk_setup *S, const plk_circuit *c, uint8_t *proof_out, uint64_t cap, uint64_t *len) {
    if (!ctx || !S || !c || !proof_out || !len) { set_error("plk_prove: bad argument"); return PLK_ERR_ARG; }
    *len = 0;
    if (!c->has_witness) { set_error("plk_prove: circuit has no witness"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    if (!ctx->srs || (!ctx->combine && ctx->srs_n < S->N)) { set_error("SRS too small for this circuit"); return PLK_ERR_SRS; }
    if (!S->store.p) { set_error("plk_prove: the setup is not on the device yet (plk_setup_upload)"); return PLK_ERR_ARG; }
    PLK_TRY(fifo_must_be_empty(ctx, "plk_prove"));
    FifoGuard fifo_guard(ctx);
    ctx->timings.clear();
    ctx->trace.valid = false;
    double t_prev = now_ms();
    auto lap = [&]() { double t = now_ms(); ctx->timings.push_back(t - t_prev); t_prev = t; };
    hipStream_t st = ctx->stream;

    // ---- witness synthesis (host): circom wires, then the transpiler's temporaries from their recorded
    //      linear forms (the gate structure itself lives in plk_setup; the reference re-synthesises here)
    if (c->r1cs.num_variables != S->num_circuit_vars || c->witness.size() < S->num_circuit_vars) {
        set_error("plk_prove: circuit does not match the prepared setup"); return PLK_ERR_ARG; }
    // circom wires are uploaded straight from the (page-locked) witness buffer; only the temporaries are
    // computed here, into a pinned staging area.  id 0 (dummy) is zeroed on the device.
    const uint64_t ncv = S->num_circuit_vars, n_tmp = S->num_vars - ncv;
    // page-locking the witness (hipHostRegister) makes its upload 2 ms faster at 2^20 but costs ~50 ms once, plus the
    // un-pinning when the process ends: worth it from the second proof of the same circuit object on, not for the
    // one-proof-per-process pattern of the CLI (profiles/r02_cli_scale.txt: 0.36 -> 0.29 s whole `plonkit prove`)
    static const int reg_mode = [] { const char *e = getenv("PLK_HOST_REGISTER"); return !e ? 1 : (!strcmp(e, "always") ? 0 : (!strcmp(e, "never") ? 1 << 30 : 1)); }();
    // Only a witness that owns its pages is page-locked: >= 4 MB sits in a 2 MiB-aligned block of its own (circuit.h, HugeAlloc).
    // A small one lives on the malloc heap and shares its 4 KiB pages with unrelated objects — numpy buffers, other circuits'
    // witnesses — that the runtime locks and unlocks on its own for pageable copies; pinning and unpinning such pages behind its
    // back ended, once in two or three runs of the whole GPU test suite, in "Memory access fault by GPU ... on address <heap page>"
    // during a LATER, unrelated host-to-device copy (round 4; profiles/r04_host_register_fault.txt).  A small upload gains nothing anyway.
    {
        std::lock_guard<std::mutex> reg_lock(c->reg_mu);
        const size_t wit_bytes = c->witness.size() * sizeof(HFr);
        if (!c->witness_registered && wit_bytes >= ((size_t)4 << 20) && (int)(c->proofs_started++) >= reg_mode) {
            if (hipHostRegister((void *)c->witness.data(), wit_bytes, hipHostRegisterDefault) == hipSuccess) c->witness_registered = true;
            else (void)hipGetLastError();                                    // not fatal: the copy is just slower
        }
    }
    static const bool tmp_host_env = [] { const char *e = getenv("PLK_WITNESS_TMP_HOST"); return e && e[0] == '1'; }();      // tests: the host loop
    const bool tmp_on_device = !tmp_host_env && (S->ops_independent || S->ops_chained) && (S->ops_dev.p || S->ops.empty());
    PLK_TRY(ensure_pinned2(ctx, (tmp_on_device ? 1 : n_tmp + 1) * sizeof(HFr)));
    HFr *tmp_vals = reinterpret_cast<HFr *>(ctx->pinned2);
    const HFr *wit = c->witness.data();
    if (!tmp_on_device) {
        const size_t n_ops = S->ops.size();
        auto value_of = [&](uint32_t v) -> HFr { return v == 0 ? HFr::zero() : (v < ncv ? wit[v] : tmp_vals[v - ncv]); };
        auto eval_range = [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) {
                const WitnessOp &op = S->ops[i];
                HFr acc = op.constant;
                for (uint32_t k = 0; k < op.count; k++) { const WitnessTerm &t = S->op_terms[op.first + k]; acc = acc + t.coeff * value_of(t.var); }
                tmp_vals[i] = acc;
            }
        };
        if (S->ops_independent && n_ops > 4096) {
            unsigned nt = std::thread::hardware_concurrency();
            if (nt > 16) nt = 16;
            if (nt < 1) nt = 1;
            std::vector<std::thread> th;
            size_t per = (n_ops + nt - 1) / nt;
            for (unsigned t = 0; t < nt; t++) { size_t lo = t * per, hi = std::min(n_ops, lo + per); if (lo < hi) th.emplace_back(eval_range, lo, hi); }
            for (auto &x : th) x.join();
        } else eval_range(0, n_ops);
    }
    struct { uint64_t num_vars; } T;
    T.num_vars = S->num_vars;
    lap();                                                                    // [0] witness synthesis

    const uint64_t N = S->N, M = 4 * N;
    const uint32_t log_n = S->log_n, log_m = log_n + 2;
    const size_t NB = (N * sizeof(Fr) + 255) & ~(size_t)255, MB = (M * sizeof(Fr) + 255) & ~(size_t)255;
    const size_t TB = ((size_t)2 * POW_TAB * sizeof(Fr) + 255) & ~(size_t)255;
    const size_t VB = (T.num_vars * sizeof(Fr) + 255) & ~(size_t)255;
    const bool direct_pi = S->num_inputs <= QUOTIENT_MAX_DIRECT_PI;       // few inputs: PI comes from the cached L0 vector inside the quotient kernel
    PLK_TRY(ctx->prove_ws.reserve(VB + 16 * NB + (direct_pi ? 6 : 7) * MB + 4 * TB + 8192));   // 5 extensions (+ PI) + the quotient
    Arena A{&ctx->prove_ws};
    Fr *d_values = A.take<Fr>(T.num_vars);
    Fr *w_vals[4], *w_coef[4];
    for (int j = 0; j < 4; j++) { w_vals[j] = A.take<Fr>(N); w_coef[j] = A.take<Fr>(N); }
    Fr *z_coef = A.take<Fr>(N), *t1 = A.take<Fr>(N), *t2 = A.take<Fr>(N), *t3 = A.take<Fr>(N);
    Fr *r_poly = A.take<Fr>(N), *agg = A.take<Fr>(N), *pi_coef = A.take<Fr>(N), *l0_coef = A.take<Fr>(N);
    Fr *ext[18] = {nullptr};
    for (int k = 0; k < 5; k++) ext[k] = A.take<Fr>(M);          // w0..w3, z
    if (!direct_pi) ext[16] = A.take<Fr>(M);                      // PI
    Fr *t_ext = A.take<Fr>(M);
    Fr *tab[4];
    for (int k = 0; k < 4; k++) tab[k] = A.take<Fr>(2 * POW_TAB);
    Fr *d_results = A.take<Fr>(16);

    uint32_t *d_flag = A.take<uint32_t>(64);
    {   // (under the circuit's lock: another context proving the SAME circuit object may be about to page-lock this buffer —
        //  not while a pageable copy of it is being staged)
        std::lock_guard<std::mutex> upload_lock(c->reg_mu);
        PLK_HIP(hipMemcpyAsync(d_values, wit, ncv * sizeof(Fr), hipMemcpyHostToDevice, st));
    }
    PLK_HIP(hipMemsetAsync(d_values, 0, sizeof(Fr), st));
    if (n_tmp && tmp_on_device && S->ops_chained)
        PLK_TRY(eval_witness_runs(d_values, S->ops_dev.p, S->terms_dev.p, S->runs_dev.p, (uint32_t)S->run_start.size(), (uint32_t)S->ops.size(), (uint32_t)ncv, st));
    else if (n_tmp && tmp_on_device) PLK_TRY(eval_witness_ops(d_values, S->ops_dev.p, S->terms_dev.p, (uint32_t)S->ops.size(), (uint32_t)ncv, st));
    else if (n_tmp) PLK_HIP(hipMemcpyAsync(d_values + ncv, tmp_vals, n_tmp * sizeof(Fr), hipMemcpyHostToDevice, st));
    std::vector<HFr> inputs(wit + 1, wit + 1 + S->num_inputs);
    {   // is_satisfied_using_one_shot_check (src/plonk.rs:137) on the device
        CheckArgs ca;
        ca.values = d_values; ca.n = (uint32_t)N; ca.num_inputs = (uint32_t)S->num_inputs; ca.flag = d_flag;
        for (int k = 0; k < 7; k++) ca.q[k] = S->sel_vals[k];
        for (int j = 0; j < 4; j++) ca.vars[j] = S->gate_vars[j];
        PLK_HIP(hipMemsetAsync(d_flag, 0, 4, st));
        PLK_TRY(check_gates(ca, st));
        // the verdict is read back into the pinned result buffer and looked at after round 1 has been enqueued (before any
        // commitment is used): the host does not stall the GPU for it.  An unsatisfied witness still ends the call with
        // PLK_ERR_UNSAT and no proof bytes — the commitments under way are drained by the FIFO guard.
        PLK_HIP(hipMemcpyAsync(ctx->pinned, d_flag, 4, hipMemcpyDeviceToHost, st));
        if (!ctx->flag_ready) PLK_HIP(hipEventCreateWithFlags(&ctx->flag_ready, hipEventDisableTiming));
        PLK_HIP(hipEventRecord(ctx->flag_ready, st));
    }

    // The extensions that no challenge waits for (wires, z, public inputs: round-3 inputs) run on a low-priority stream of
    // their own with their own NTT scratch: they fill the SIMDs that the latency-bound ends of a commitment leave idle
    // without standing between two rounds — on the main stream round 2 queued up behind the four wire extensions, which
    // in turn were starved by the accumulation they shared the GPU with (round 1: 6.4-7.0 ms).  Up to the 2^24 domain (a
    // second 4N scratch above that is memory better spent elsewhere); PLK_PROVE_BG=0 restores the single stream.
    static const bool bg_env = [] { const char *e = getenv("PLK_PROVE_BG"); return !(e && e[0] == '0'); }();
    const bool use_bg = bg_env && log_n <= 24;
    if (use_bg && !ctx->bg_stream) {
        int lo_prio = 0, hi_prio = 0;
        PLK_HIP(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
        PLK_HIP(hipStreamCreateWithPriority(&ctx->bg_stream, hipStreamNonBlocking, lo_prio));
        PLK_HIP(hipEventCreateWithFlags(&ctx->bg_go, hipEventDisableTiming));
        PLK_HIP(hipEventCreateWithFlags(&ctx->bg_done, hipEventDisableTiming));
    }
    hipStream_t bg = use_bg ? ctx->bg_stream : st;
    const uint32_t bg_lane = use_bg ? 1 : 0;
    struct BgGuard {                                          // an early return must not leave background work writing into the arena
        hipStream_t s; bool armed;
        ~BgGuard() { if (armed && s) (void)hipStreamSynchronize(s); }
    } bg_guard{use_bg ? ctx->bg_stream : nullptr, use_bg};
    // What follows on bg starts after everything enqueued on st so far AND after the accumulation of the commitment enqueued
    // last: stream priorities do not keep a 8192-workgroup NTT pass from taking the CUs first (measured: the pre-phase of the
    // wire commitments took 2.1 ms instead of 0.35 behind it), so the start is placed by hand where the GPU has room — the
    // bucket reduction of that commitment, the point-wise kernels of the next round and the pre-phase of its commitment.
    static const int bg_gate_z = [] { const char *e = getenv("PLK_PROVE_BG_GATE_Z"); return e ? atoi(e) : 0; }();   // A/B knob
    static const int bg_gate_w = [] { const char *e = getenv("PLK_PROVE_BG_GATE_W"); return e ? atoi(e) : 1; }();   // A/B knob: wire extensions behind the wires' accumulation
    auto bg_after_main = [&](bool behind_accumulation) -> int32_t {
        if (!use_bg) return PLK_OK;
        PLK_HIP(hipEventRecord(ctx->bg_go, st));
        PLK_HIP(hipStreamWaitEvent(bg, ctx->bg_go, 0));
        if (behind_accumulation && ctx->msm_enq != ctx->msm_fin) {
            plk_ctx::MsmSlot &L = ctx->slot[ctx->fifo[(ctx->msm_enq - 1) % plk_ctx::MSM_SLOTS]];
            if (L.acc_done) PLK_HIP(hipStreamWaitEvent(bg, L.acc_done, 0));
        }
        return PLK_OK;
    };

    // ---- round 1: wire polynomials, 4 x iNTT(N) in one launch per pass, 4 x MSM(N)
    PLK_TRY(gather4_dual(w_vals, w_coef, d_values, S->gate_vars, (uint32_t)N, st));
    if (log_n <= 22) PLK_TRY(ntt_batch_dev(ctx, w_coef, 4, log_n, true, nullptr, st, 0));
    else for (int j = 0; j < 4; j++) PLK_TRY(ntt_dev(ctx, w_coef[j], log_n, true, nullptr, st));
    // with a Lagrange-form key of the domain's size resident (`prove -l`, src/plonk.rs:138-146) the witness and
    // grand-product polynomials are committed from their evaluations, as bellman's prove() does; same proof bytes
    const bool use_lagrange = ctx->lag.pts != nullptr;
    if (use_lagrange && (ctx->combine ? ctx->lag.n != ctx->srs_n : ctx->lag.n != N)) { set_error("Lagrange-form key has a different size than the circuit's domain"); return PLK_ERR_SRS; }
    HAffine wire_c[4];
    PLK_TRY(commit_begin(ctx, use_lagrange ? w_vals : w_coef, 4, N, use_lagrange));
    PLK_TRY(bg_after_main(bg_gate_w != 0));
    PLK_TRY(lde4cm_batch_dev(ctx, w_coef, 4, log_n, ext, bg, bg_lane));                       // round-3 work that needs no challenge
    PLK_HIP(hipEventSynchronize(ctx->flag_ready));
    if (*reinterpret_cast<volatile uint32_t *>(ctx->pinned)) { set_error("must satisfy: witness does not satisfy the circuit"); return PLK_ERR_UNSAT; }
    PLK_TRY(commit_end(ctx, 4, wire_c));
    RollingKeccak tr;
    for (const HFr &x : inputs) tr.absorb_fr(x);
    for (int j = 0; j < 4; j++) tr.absorb_g1(wire_c[j]);
    const HFr beta = tr.challenge(), gamma = tr.challenge();
    lap();                                                                    // [1] round 1

    // ---- round 2: grand product z  (z_i = prod_{k<i} num_k / den_k, no per-element inversion:
    //      z_i = A_i * C_i / C_0 with A = exclusive prefix product of num, C = inclusive suffix product of den)
    HFr kk[4];
    for (int j = 0; j < 4; j++) kk[j] = HFr::from_u64(NON_RESIDUES[j]);
    {
        PermArgs pa;
        pa.num = t1; pa.den = t2;
        for (int j = 0; j < 4; j++) { pa.w[j] = w_vals[j]; pa.sigma[j] = S->sig_vals[j]; pa.beta_k[j] = to_dev(beta * kk[j]); }
        pa.beta = to_dev(beta * HFr::from_u64(32)); pa.gamma = to_dev(gamma); pa.fix = to_dev(HFr::from_u64(1u << 25));   // domains: poly.h
        pa.n = (uint32_t)N; pa.log_n = log_n; pa.tw = ctx->tw_fwd_w;
        PLK_TRY(perm_terms(pa, st));
        // (the scans stop after their second phase when they span several blocks: the block prefixes are folded in by the product below,
        //  and C_0 — the product of all denominators — is the suffix scan's grand total, which its second phase leaves behind the prefixes)
        const Fr *pre_a = nullptr, *pre_c = nullptr;
        static const bool fuse_scan_tail = [] { const char *e = getenv("PLK_PROVE_FUSE_SCAN_TAIL"); return !(e && e[0] == '0'); }();   // A/B knob, read once
        PLK_TRY(scan_pair_mult(ctx, t1, t1, false, true, t2, t2, true, false, (uint32_t)N, st, fuse_scan_tail ? &pre_a : nullptr, fuse_scan_tail ? &pre_c : nullptr));
        const uint32_t scan_blocks = pre_c ? (uint32_t)((N + POLY_SCAN_BLOCK - 1) / POLY_SCAN_BLOCK) : 0;
        HFr total;
        PLK_HIP(hipMemcpyAsync(total.l, pre_c ? pre_c + scan_blocks : t2, sizeof(Fr), hipMemcpyDeviceToHost, st));
        PLK_HIP(hipStreamSynchronize(st));
        if (total.is_zero()) { set_error("grand product denominator vanished (probability ~2^-230)"); return PLK_ERR_UNSAT; }
        // the scans live in the W domain: what was read is 32 * C_0, so E(1 / C_0) = 32 * E(1 / (32 C_0))
        const Fr inv_c0 = to_dev(total.inv() * HFr::from_u64(32));
        if (pre_c) PLK_TRY(mul3_blocks(z_coef, t1, t2, pre_a, pre_c, inv_c0, (uint32_t)N, st));
        else PLK_TRY(mul3(z_coef, t1, t2, inv_c0, (uint32_t)N, st));
        if (use_lagrange) PLK_HIP(hipMemcpyAsync(t1, z_coef, N * sizeof(Fr), hipMemcpyDeviceToDevice, st));   // keep the values
        PLK_TRY(ntt_dev(ctx, z_coef, log_n, true, nullptr, st));
    }
    HAffine z_c;
    { const Fr *zp = use_lagrange ? t1 : z_coef; PLK_TRY(commit_begin(ctx, &zp, 1, N, use_lagrange)); }
    // while z is being committed: its extension, the public-input polynomial, and (first proof only) the constant vectors
    // (z's extension is the last thing the quotient waits for: it starts as soon as z's coefficients exist and shares the GPU
    //  with the accumulation of z's commitment — behind that accumulation it ended 0.4 ms after the commitment itself)
    PLK_TRY(bg_after_main(bg_gate_z != 0));
    {
        const Fr *zc = z_coef;
        PLK_TRY(lde4cm_batch_dev(ctx, &zc, 1, log_n, &ext[4], bg, bg_lane));
    }
    if (!direct_pi) {
        PLK_HIP(hipMemsetAsync(pi_coef, 0, N * sizeof(Fr), bg));
        PLK_HIP(hipMemcpyAsync(pi_coef, inputs.data(), inputs.size() * sizeof(Fr), hipMemcpyHostToDevice, bg));
        PLK_TRY(ntt_batch_dev(ctx, &pi_coef, 1, log_n, true, nullptr, bg, bg_lane));
        const Fr *pc = pi_coef;
        PLK_TRY(lde4cm_batch_dev(ctx, &pc, 1, log_n, &ext[16], bg, bg_lane));
    }
    if (use_bg) PLK_HIP(hipEventRecord(ctx->bg_done, bg));
    PLK_TRY(commit_end(ctx, 1, &z_c));
    tr.absorb_g1(z_c);
    const HFr alpha = tr.challenge();
    lap();                                                                    // [2] round 2

    // ---- round 3: quotient on the coset 7*<omega_4N>: 18 x LDE, fused point-wise kernel, coset iNTT(4N)
    const HFr coset = HFr::from_u64(7);
    {
        std::unique_lock<std::mutex> lazy_lock(S->lazy_mu);
        if (!S->lde_ready) {
            // the coset-point vector is a convenience (one load instead of two loads and two products per point):
            // above 2^24 gates its 4N * 32 bytes are better spent elsewhere (2^26 would not fit in 288 GB)
            static const bool no_cache_env = getenv("PLK_NO_COSET_CACHE") != nullptr;       // tests: the large-domain branch at a small size
            const bool cache_x = log_n <= 24 && !no_cache_env;
            PLK_TRY(S->lde_store.reserve((cache_x ? 13 : 12) * MB));
            Arena LA{&S->lde_store};
            for (int k = 0; k < (cache_x ? 13 : 12); k++) S->lde[k] = LA.take<Fr>(M);
            if (!cache_x) S->lde[12] = nullptr;
            HFr one = HFr::one();
            PLK_HIP(hipMemsetAsync(l0_coef, 0, N * sizeof(Fr), st));
            PLK_HIP(hipMemcpyAsync(l0_coef, one.l, sizeof(Fr), hipMemcpyHostToDevice, st));
            PLK_TRY(ntt_dev(ctx, l0_coef, log_n, true, nullptr, st));
            {   // 7 selectors, 4 sigmas, L0: twelve extensions in the coset-major layout of the quotient kernel, four polynomials per launch
                const Fr *src[12]; Fr *dst[12];
                for (int k = 0; k < 7; k++) { src[k] = S->sel_coef[k]; dst[k] = S->lde[k]; }
                for (int j = 0; j < 4; j++) { src[7 + j] = S->sig_coef[j]; dst[7 + j] = S->lde[7 + j]; }
                src[11] = l0_coef; dst[11] = S->lde[11];
                PLK_TRY(lde4cm_batch_dev(ctx, src, 12, log_n, dst, st, 0));
            }
            // pre-scaled for the 29-bit layer of the quotient kernel (poly.h, QuotientArgs): 2^5 everywhere,
            // 2^10 on q_m, none on q_const (it is only added); plus the coset points x_i = 7 * omega_4N^i
            const HFr s5 = HFr::from_u64(1u << 10), s10 = HFr::from_u64(1u << 15);      // as factors of a product that removes 2^5
            for (int k = 0; k < 12; k++) {
                if (k == 5) continue;
                PLK_TRY(scale_const(S->lde[k], S->lde[k], to_dev(k == 4 ? s10 : s5), (uint32_t)M, st));
            }
            if (S->lde[12]) PLK_TRY(coset_points_w(S->lde[12], ctx->tw_fwd_w, log_m, to_dev(HFr::from_u64(7u << 5)), (uint32_t)M, st));
            PLK_HIP(hipStreamSynchronize(st));
            S->lde_ready = true;
        }
        for (int k = 0; k < 11; k++) ext[5 + k] = S->lde[k];
        ext[17] = S->lde[11];
        QuotientArgs qa;
        qa.out = t_ext;
        for (int j = 0; j < 4; j++) { qa.w[j] = ext[j]; qa.sigma[j] = ext[12 + j]; qa.beta_k[j] = to_dev(beta * kk[j]); }
        qa.z = ext[4];
        for (int k = 0; k < 7; k++) qa.q[k] = ext[5 + k];
        qa.pi = direct_pi ? nullptr : ext[16]; qa.l0 = ext[17]; qa.x = S->lde[12];
        qa.tw_w = ctx->tw_fwd_w; qa.coset_w = to_dev(HFr::from_u64(7u << 5));
        qa.num_pi = direct_pi ? (uint32_t)inputs.size() : 0;
        for (uint32_t k = 0; k < qa.num_pi; k++) qa.pi_in[k] = to_dev(inputs[k]);
        const HFr two5 = HFr::from_u64(1u << 5), two25 = HFr::from_u64(1u << 25);
        qa.beta = to_dev(beta); qa.gamma = to_dev(gamma);
        qa.alpha_pp = to_dev(alpha * two25); qa.alpha2_w = to_dev(alpha * alpha * two5);
        if (!S->zh_inv_ready) {                                   // 1 / Z_H on the four cosets of <omega_N> inside the 4N domain: circuit constants
            HFr gN = coset.pow_u64(N), iota = host_omega(log_m).pow_u64(N), ip = HFr::one();
            for (int k = 0; k < 4; k++) { S->zh_inv[k] = (gN * ip - HFr::one()).inv(); ip = ip * iota; }
            const HFr gN_inv = gN.inv(), quarter = HFr::from_u64(4).inv();
            S->icoset_c[0] = iota.inv();                           // iota = omega_4N^N = omega_4
            S->icoset_c[1] = quarter;
            for (int c = 1; c < 4; c++) S->icoset_c[1 + c] = S->icoset_c[c] * gN_inv;
            S->zh_inv_ready = true;
        }
        lazy_lock.unlock();
        for (int k = 0; k < 4; k++) qa.zh_inv_w[k] = to_dev(S->zh_inv[k] * two5);
        qa.m = (uint32_t)M; qa.log_m = log_m;
        if (use_bg) PLK_HIP(hipStreamWaitEvent(st, ctx->bg_done, 0));      // the five extensions (+ PI) of the background stream
        PLK_TRY(quotient(qa, st));
        // coset iNTT at 4N from the coset-major layout the kernel wrote: four inverse N-point coset transforms (one launch per
        // pass) and a 4-point combine, instead of three passes over 4N (0.57 -> 0.48 ms at the 2^20 domain)
        PLK_TRY(icoset4cm_dev(ctx, t_ext, log_n, st, 0));
        {
            const HFr two5 = HFr::from_u64(32);
            Fr s_w[4];
            for (int c = 0; c < 4; c++) s_w[c] = to_dev(S->icoset_c[1 + c] * two5);
            PLK_TRY(icoset_combine(t_ext, (uint32_t)N, to_dev(S->icoset_c[0] * two5), s_w, st));
        }
    }
    HAffine t_c[4];
    {
        const Fr *parts[4] = {t_ext, t_ext + N, t_ext + 2 * N, t_ext + 3 * N};
        PLK_TRY(commit_many(ctx, parts, 4, N, t_c));
    }
    bg_guard.armed = false;                                   // the quotient consumed everything the background stream produced
    for (int k = 0; k < 4; k++) tr.absorb_g1(t_c[k]);
    const HFr z = tr.challenge();
    lap();                                                                    // [3] round 3

    // ---- round 4: evaluations at z and z*omega, linearisation
    const HFr omega = host_omega(log_n), zw = z * omega, zN = z.pow_u64(N);
    if (z.is_zero()) { set_error("challenge z = 0"); return PLK_ERR_UNSAT; }
    // the three inversions of this round (1/z, 1/(z omega), 1/(N (z - 1)) for L_0(z)) share one: the host inversion is what the
    // GPU waits for between the rounds
    const HFr l0_den = HFr::from_u64(N) * (z - HFr::one());
    if (l0_den.is_zero()) { set_error("challenge z = 1"); return PLK_ERR_UNSAT; }
    const HFr inv_all = (z * zw * l0_den).inv();
    const HFr z_inv = inv_all * zw * l0_den, zw_inv = inv_all * z * l0_den, l0_den_inv = inv_all * z * zw;
    PowTable pts4[4];
    {
        const Fr bases4[4] = {to_dev(z), to_dev(z_inv), to_dev(zw), to_dev(zw_inv)};
        PLK_TRY(fill_pow_tables4_into(ctx, bases4, tab, pts4, st));
    }
    const PowTable pt_z = pts4[0], pt_zinv = pts4[1], pt_zw = pts4[2], pt_zwinv = pts4[3];
    HFr ev[11];
    {
        EvalArgs ea{};
        const Fr *polys[10] = {w_coef[0], w_coef[1], w_coef[2], w_coef[3], w_coef[3], S->sig_coef[0], S->sig_coef[1], S->sig_coef[2], t_ext, z_coef};
        for (int e = 0; e < 10; e++) { ea.poly[e] = polys[e]; ea.len[e] = (uint32_t)(e == 8 ? M : N); ea.pt[e] = (e == 4 || e == 9) ? pt_zw : pt_z; }
        ea.count = 10;
        PLK_TRY(eval_batch(ctx, ea, d_results, st));
        PLK_HIP(hipMemcpyAsync(ev, d_results, 10 * sizeof(Fr), hipMemcpyDeviceToHost, st));
        PLK_HIP(hipStreamSynchronize(st));
    }
    const HFr *wz = ev, w3zw = ev[4], *sz = ev + 5, tz = ev[8], zzw = ev[9];
    HFr l0z = (zN - HFr::one()) * l0_den_inv;
    HFr fz = alpha;
    for (int j = 0; j < 4; j++) fz = fz * (wz[j] + beta * kk[j] * z + gamma);
    fz = fz + alpha * alpha * l0z;
    HFr fs = alpha * beta * zzw;
    for (int j = 0; j < 3; j++) fs = fs * (wz[j] + beta * sz[j] + gamma);
    {
        LinCombArgs la{};
        la.out = r_poly; la.n = (uint32_t)N; la.count = 9;
        const Fr *ps[9] = {S->sel_coef[5], S->sel_coef[0], S->sel_coef[1], S->sel_coef[2], S->sel_coef[3], S->sel_coef[4], S->sel_coef[6], z_coef, S->sig_coef[3]};
        HFr sc[9] = {HFr::one(), wz[0], wz[1], wz[2], wz[3], wz[0] * wz[1], w3zw, fz, -fs};
        for (int k = 0; k < 9; k++) { la.p[k] = ps[k]; la.s[k] = to_dev(sc[k] * HFr::from_u64(32)); la.unit[k] = (k == 0); }
        PLK_TRY(lincomb(la, st));
        EvalArgs ea{};
        ea.poly[0] = r_poly; ea.len[0] = (uint32_t)N; ea.pt[0] = pt_z; ea.count = 1;
        PLK_TRY(eval_batch(ctx, ea, d_results, st));
        PLK_HIP(hipMemcpyAsync(&ev[10], d_results, sizeof(Fr), hipMemcpyDeviceToHost, st));
        PLK_HIP(hipStreamSynchronize(st));
    }
    const HFr rz = ev[10];
    for (int j = 0; j < 4; j++) tr.absorb_fr(wz[j]);
    tr.absorb_fr(w3zw);
    for (int j = 0; j < 3; j++) tr.absorb_fr(sz[j]);
    tr.absorb_fr(tz); tr.absorb_fr(rz); tr.absorb_fr(zzw);
    const HFr v = tr.challenge();
    lap();                                                                    // [4] round 4

    // ---- round 5: opening proofs W_z, W_zw by synthetic division (suffix sums of p_j z^j, times z^-(k+1))
    HAffine Wz, Wzw;
    {
        LinCombArgs la{};
        la.out = agg; la.n = (uint32_t)N; la.count = 12;
        HFr vp[11]; vp[0] = HFr::one();
        for (int k = 1; k <= 10; k++) vp[k] = vp[k - 1] * v;
        const Fr *ps[12] = {t_ext, t_ext + N, t_ext + 2 * N, t_ext + 3 * N, r_poly, w_coef[0], w_coef[1], w_coef[2], w_coef[3],
                            S->sig_coef[0], S->sig_coef[1], S->sig_coef[2]};
        HFr sc[12] = {HFr::one(), zN, zN * zN, zN * zN * zN, vp[1], vp[2], vp[3], vp[4], vp[5], vp[6], vp[7], vp[8]};
        for (int k = 0; k < 12; k++) { la.p[k] = ps[k]; la.s[k] = to_dev(sc[k] * HFr::from_u64(32)); la.unit[k] = (k == 0); }
        // The two quotients are independent chains of five small launches each (linear combination times z^i, suffix sums,
        // times z^-(k+1)): the second one runs on the background stream (idle since round 3) with scratch of its own, so that
        // the two chains overlap instead of queueing (0.14 ms of pure launch latency at the 2^20 domain).
        la.times_pow = pt_z; la.out = t1;
        LinCombArgs lb{};
        lb.n = (uint32_t)N; lb.count = 2;
        lb.p[0] = z_coef; lb.s[0] = to_dev(vp[9] * HFr::from_u64(32)); lb.p[1] = w_coef[3]; lb.s[1] = to_dev(vp[10] * HFr::from_u64(32));
        lb.times_pow = pt_zw;
        Fr *const b_tmp = use_bg ? pi_coef : t1;                      // (pi_coef: free since the quotient)
        hipStream_t sb = use_bg ? bg : st;
        lb.out = b_tmp;
        bg_guard.armed = use_bg;                                      // (an error below must not leave chain B writing into the arena)
        if (use_bg) { PLK_HIP(hipEventRecord(ctx->bg_go, st)); PLK_HIP(hipStreamWaitEvent(sb, ctx->bg_go, 0)); }   // the power tables and r(x) come from st
        PLK_TRY(lincomb(la, st));
        PLK_TRY(scan(ctx, t1, t1, (uint32_t)N, false, true, false, st));
        PLK_TRY(div_finish(t2, t1, pt_zinv, (uint32_t)N, st));
        PLK_TRY(lincomb(lb, sb));
        PLK_TRY(scan(ctx, b_tmp, b_tmp, (uint32_t)N, false, true, false, sb, use_bg ? &ctx->poly_tmp2 : nullptr));
        PLK_TRY(div_finish(t3, b_tmp, pt_zwinv, (uint32_t)N, sb));
        if (use_bg) { PLK_HIP(hipEventRecord(ctx->bg_done, sb)); PLK_HIP(hipStreamWaitEvent(st, ctx->bg_done, 0)); }
        const Fr *opens[2] = {t2, t3};
        HAffine oc[2];
        PLK_TRY(commit_many(ctx, opens, 2, N, oc));
        bg_guard.armed = false;                                       // the commitments consumed t3: chain B has ended
        Wz = oc[0]; Wzw = oc[1];
    }
    lap();                                                                    // [5] round 5

    // ---- Proof::write
    std::vector<uint8_t> b;
    put_u64(b, S->n);
    put_u64(b, inputs.size());
    for (const HFr &x : inputs) put_fr(b, x);
    put_u64(b, 4); for (int j = 0; j < 4; j++) put_g1(b, wire_c[j]);
    put_g1(b, z_c);
    put_u64(b, 4); for (int k = 0; k < 4; k++) put_g1(b, t_c[k]);
    put_u64(b, 4); for (int j = 0; j < 4; j++) put_fr(b, wz[j]);
    put_u64(b, 1); put_fr(b, w3zw);
    put_fr(b, zzw); put_fr(b, tz); put_fr(b, rz);
    put_u64(b, 3); for (int j = 0; j < 3; j++) put_fr(b, sz[j]);
    put_g1(b, Wz); put_g1(b, Wzw);
    *len = b.size();
    {   // what plk_prove_trace hands out (all of it lives in ctx->prove_ws until the next proof)
        plk_ctx::Trace &tr_ = ctx->trace;
        for (int j = 0; j < 4; j++) { tr_.ptr[j] = w_coef[j]; tr_.len[j] = N; }
        tr_.ptr[4] = z_coef; tr_.len[4] = N;
        tr_.ptr[5] = t_ext; tr_.len[5] = M;
        tr_.ptr[6] = r_poly; tr_.len[6] = N;
        tr_.ptr[7] = t2; tr_.len[7] = N;
        tr_.ptr[8] = t3; tr_.len[8] = N;
        tr_.valid = true;
    }
    if (b.size() > cap) { set_error("plk_prove: proof buffer too small"); return PLK_ERR_ARG; }
    memcpy(proof_out, b.data(), b.size());
    lap();                                                                    // [6] serialise
    return PLK_OK;
}

// ---- tracing / test hooks: the vectors between the rounds, and the polynomial helpers of rounds 2, 4 and 5 on their own
int32_t plk_prove_trace(plk_ctx *ctx, uint32_t which, plk_fr *out_host, uint64_t cap, uint64_t *n) {
    if (!ctx || !n || which >= 9) { set_error("plk_prove_trace: bad argument"); return PLK_ERR_ARG; }
    if (!ctx->trace.valid) { set_error("plk_prove_trace: no finished plk_prove on this context"); return PLK_ERR_ARG; }
    *n = ctx->trace.len[which];
    if (!out_host) return PLK_OK;
    if (cap < *n) { set_error("plk_prove_trace: buffer too small"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_HIP(hipMemcpyAsync(out_host, ctx->trace.ptr[which], *n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return PLK_OK;
}

int32_t plk_poly_evaluate_at_dev(plk_ctx *ctx, const void *coeffs_dev, uint64_t n, const plk_fr *z, plk_fr *out, void *stream) {
    if (!ctx || !coeffs_dev || !z || !out || n == 0 || n > (1ull << MAX_LOG_N)) { set_error("plk_poly_evaluate_at_dev: bad argument"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    DevBuf tab;
    int32_t rc = tab.reserve((size_t)2 * POW_TAB * sizeof(Fr) + 64);
    if (rc != PLK_OK) return rc;
    Fr zz; memcpy(zz.l, z->l, 32);
    PowTable pt;
    rc = fill_pow_table_into(ctx, zz, tab.as<Fr>(), &pt, st);
    EvalArgs ea{};
    ea.poly[0] = (const Fr *)coeffs_dev; ea.len[0] = (uint32_t)n; ea.pt[0] = pt; ea.count = 1;
    Fr *res = tab.as<Fr>() + 2 * POW_TAB;
    if (rc == PLK_OK) rc = eval_batch(ctx, ea, res, st);
    if (rc == PLK_OK && hipMemcpyAsync(out, res, sizeof(Fr), hipMemcpyDeviceToHost, st) != hipSuccess) rc = hip_fail(hipGetLastError(), "D2H", __FILE__, __LINE__);
    if (hipStreamSynchronize(st) != hipSuccess && rc == PLK_OK) rc = hip_fail(hipGetLastError(), "sync", __FILE__, __LINE__);
    tab.release();
    return rc;
}

int32_t plk_poly_divide_by_linear_dev(plk_ctx *ctx, const void *coeffs_dev, uint64_t n, const plk_fr *z, void *quotient_dev, void *stream) {
    if (!ctx || !coeffs_dev || !z || !quotient_dev || n == 0 || n > (1ull << MAX_LOG_N)) { set_error("plk_poly_divide_by_linear_dev: bad argument"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    HFr hz; memcpy(hz.l, z->l, 32);
    if (hz.is_zero()) { set_error("plk_poly_divide_by_linear_dev: z = 0"); return PLK_ERR_ARG; }
    DevBuf tab;
    int32_t rc = tab.reserve((size_t)4 * POW_TAB * sizeof(Fr) + (size_t)n * sizeof(Fr));
    if (rc != PLK_OK) return rc;
    PowTable pz, pzi;
    Fr *tmp = tab.as<Fr>() + 4 * POW_TAB;
    rc = fill_pow_table_into(ctx, to_dev(hz), tab.as<Fr>(), &pz, st);
    if (rc == PLK_OK) rc = fill_pow_table_into(ctx, to_dev(hz.inv()), tab.as<Fr>() + 2 * POW_TAB, &pzi, st);
    // (p(x) - p(z)) / (x - z): q_k = z^-(k+1) * sum_{j > k} p_j z^j  — the schedule of round 5
    if (rc == PLK_OK) rc = mul_powers(tmp, (const Fr *)coeffs_dev, pz, 0, (uint32_t)n, st);
    if (rc == PLK_OK) rc = scan(ctx, tmp, tmp, (uint32_t)n, false, true, false, st);
    if (rc == PLK_OK) rc = div_finish((Fr *)quotient_dev, tmp, pzi, (uint32_t)n, st);
    if (hipStreamSynchronize(st) != hipSuccess && rc == PLK_OK) rc = hip_fail(hipGetLastError(), "sync", __FILE__, __LINE__);
    tab.release();
    return rc;
}

int32_t plk_permutation_grand_product_dev(plk_ctx *ctx, const void *const wires_dev[4], const void *const sigmas_dev[4], const plk_fr *beta, const plk_fr *gamma,
                                          uint32_t log_n, void *z_values_dev, void *stream) {
    if (!ctx || !wires_dev || !sigmas_dev || !beta || !gamma || !z_values_dev || log_n > MAX_LOG_N) { set_error("plk_permutation_grand_product_dev: bad argument"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    const uint64_t N = 1ull << log_n;
    DevBuf tmp;
    PLK_TRY(tmp.reserve(2 * N * sizeof(Fr)));
    HFr hb, hg; memcpy(hb.l, beta->l, 32); memcpy(hg.l, gamma->l, 32);
    PermArgs pa;
    pa.num = tmp.as<Fr>(); pa.den = tmp.as<Fr>() + N;
    for (int j = 0; j < 4; j++) { pa.w[j] = (const Fr *)wires_dev[j]; pa.sigma[j] = (const Fr *)sigmas_dev[j]; pa.beta_k[j] = to_dev(hb * HFr::from_u64(NON_RESIDUES[j])); }
    pa.beta = to_dev(hb * HFr::from_u64(32)); pa.gamma = to_dev(hg); pa.fix = to_dev(HFr::from_u64(1u << 25));
    pa.n = (uint32_t)N; pa.log_n = log_n; pa.tw = ctx->tw_fwd_w;
    int32_t rc = perm_terms(pa, st);
    if (rc == PLK_OK) rc = scan_pair_mult(ctx, pa.num, pa.num, false, true, pa.den, pa.den, true, false, (uint32_t)N, st);
    HFr total;
    if (rc == PLK_OK && hipMemcpyAsync(total.l, pa.den, sizeof(Fr), hipMemcpyDeviceToHost, st) != hipSuccess) rc = hip_fail(hipGetLastError(), "D2H", __FILE__, __LINE__);
    if (rc == PLK_OK && hipStreamSynchronize(st) != hipSuccess) rc = hip_fail(hipGetLastError(), "sync", __FILE__, __LINE__);
    if (rc == PLK_OK && total.is_zero()) { set_error("grand product denominator vanished"); rc = PLK_ERR_UNSAT; }
    if (rc == PLK_OK) rc = mul3((Fr *)z_values_dev, pa.num, pa.den, to_dev(total.inv() * HFr::from_u64(32)), (uint32_t)N, st);
    if (hipStreamSynchronize(st) != hipSuccess && rc == PLK_OK) rc = hip_fail(hipGetLastError(), "sync", __FILE__, __LINE__);
    tmp.release();
    return rc;
}

// the exported entry points: no C++ exception (std::bad_alloc from a host vector of a 2^26 domain) crosses the boundary
int32_t plk_setup_prepare(plk_ctx *ctx, const plk_circuit *c, plk_setup **out) {
    return guarded("plk_setup_prepare", PLK_ERR_HIP, [&] { return setup_prepare_impl(ctx, c, out); });
}
int32_t plk_setup_prepare_host(const plk_circuit *c, plk_setup **out) {
    return guarded("plk_setup_prepare_host", PLK_ERR_FORMAT, [&] { return setup_host_impl(c, out); });
}
int32_t plk_setup_upload(plk_ctx *ctx, plk_setup *s) {
    return guarded("plk_setup_upload", PLK_ERR_HIP, [&] { return setup_upload_impl(ctx, s); });
}
static int32_t not_a_worker(plk_ctx *ctx, const char *who) {
    if (!ctx || !comm_scatter_worker(ctx)) return PLK_OK;
    set_error(std::string(who) + ": this rank serves the owner's commitments (owner-computes mode: call plk_comm_serve; only rank 0 proves)");
    return PLK_ERR_ARG;
}
int32_t plk_setup_write_vk(plk_ctx *ctx, const plk_setup *s, const uint8_t g2_bytes[256], uint8_t *out, uint64_t cap, uint64_t *len) {
    PLK_TRY(not_a_worker(ctx, "plk_setup_write_vk"));
    return guarded("plk_setup_write_vk", PLK_ERR_HIP, [&] { return setup_write_vk_impl(ctx, s, g2_bytes, out, cap, len); });
}
int32_t plk_prove(plk_ctx *ctx, const plk_setup *S, const plk_circuit *c, uint8_t *proof_out, uint64_t cap, uint64_t *len) {
    PLK_TRY(not_a_worker(ctx, "plk_prove"));
    return guarded("plk_prove", PLK_ERR_HIP, [&] { return prove_impl(ctx, S, c, proof_out, cap, len); });
}
int32_t plk_comm_serve(plk_ctx *ctx, uint64_t *batches) {
    if (batches) *batches = 0;
    if (!ctx || !comm_scatter_worker(ctx)) { set_error("plk_comm_serve: not a worker rank (rank > 0) of a communicator in scatter mode"); return PLK_ERR_ARG; }
    if (!ctx->srs) { set_error("plk_comm_serve: no key slice resident"); return PLK_ERR_SRS; }
    PLK_TRY(fifo_must_be_empty(ctx, "plk_comm_serve"));
    return guarded("plk_comm_serve", PLK_ERR_HIP, [&] { PLK_HIP(hipSetDevice(ctx->device)); FifoGuard drain(ctx); return serve_impl(ctx, batches); });
}

}  // extern "C"
