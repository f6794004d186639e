// Context lifetime, error reporting, SRS residency and the host-pointer wrappers of the C ABI.
// (The reference's counterpart is bellman_ce::worker::Worker + Crs held in SetupForProver,
//  src/plonk.rs:41-55; here the resource is one MI355X, its stream and its HBM-resident tables.)
#include "ctx.h"
#include <cstdlib>
#include "ntt.h"
#include "poly.h"
#include "hostmath.h"
#include "msm.h"
#include "comm.h"
#include <cstring>
#include <cstdio>
#include <mutex>

namespace plk {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int32_t hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    g_last_error = buf;
    (void)hipGetLastError();
    return PLK_ERR_HIP;
}

int32_t DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return PLK_OK;
    if (borrowed) { set_error("a buffer borrowed from another context (plk_ctx_share_srs) cannot grow"); return PLK_ERR_ARG; }
    if (p) { PLK_HIP(hipFree(p)); p = nullptr; cap = 0; }
    size_t slack = bytes >> 3;                    // a little slack so that growth is rare ...
    if (slack > ((size_t)64 << 20)) slack = (size_t)64 << 20;     // ... but never 12 % of a 100 GB workspace (2^26 domains)
    size_t want = bytes + slack;
    PLK_HIP(hipMalloc(&p, want));
    cap = want;
    return PLK_OK;
}

void DevBuf::release() {
    if (p && !borrowed) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    borrowed = false;
}

int32_t ensure_pinned(plk_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->pinned_cap) return PLK_OK;
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = nullptr;
    PLK_HIP(hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault));
    ctx->pinned_cap = bytes;
    return PLK_OK;
}

int32_t ensure_pinned2(plk_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->pinned2_cap) return PLK_OK;
    if (ctx->pinned2) (void)hipHostFree(ctx->pinned2);
    ctx->pinned2 = nullptr; ctx->pinned2_cap = 0;
    size_t want = bytes + (bytes >> 2);
    PLK_HIP(hipHostMalloc(&ctx->pinned2, want, hipHostMallocDefault));
    ctx->pinned2_cap = want;
    return PLK_OK;
}

// a context whose key is on loan must keep it: the borrowers' commitments read it (and its MSM table) at any time.  The Lagrange-form
// key of a lender is only frozen while a borrower actually holds it (a lender that had none when the loans were made may install one;
// its borrowers do not see it — share again for that).
static std::mutex g_share_mu;                          // loans are made, returned and orphaned under one lock (rare operations)
int32_t srs_replace_guard(plk_ctx *c, const char *who, bool lagrange_only) {
    if (lagrange_only ? c->lag_borrowers.load() > 0 : c->srs_borrowers.load() > 0) {
        set_error(std::string(who) + ": this context's key is shared with another context (plk_ctx_share_srs) — destroy the borrowers first");
        return PLK_ERR_ARG;
    }
    if (!lagrange_only) srs_return_loan(c);              // a borrower that gets a monomial key of its own stops borrowing altogether;
    else if (c->lag_borrowed) {                          // one that replaces its Lagrange-form key keeps the borrowed monomial key
        std::lock_guard<std::mutex> g(g_share_mu);
        c->srs_lender->lag_borrowers.fetch_sub(1);
        c->lag_borrowed = false;
        c->lag.pts = nullptr; c->lag.n = 0; c->lag.w.release(); c->lag.w_valid = false; c->lag.w_copies = 0;
    }
    return PLK_OK;
}
static void free_lent(plk_ctx *c) {                      // what a destroyed lender had to keep for its borrowers
    (void)hipSetDevice(c->device);
    c->srs_own.release(); c->srs_w.release(); c->lag.own.release(); c->lag.w.release();
    delete c;
}
void srs_return_loan(plk_ctx *c) {
    if (!c->srs_lender) return;
    plk_ctx *orphan = nullptr;
    {
        std::lock_guard<std::mutex> g(g_share_mu);
        plk_ctx *const lender = c->srs_lender;
        if (c->lag_borrowed) lender->lag_borrowers.fetch_sub(1);
        if (lender->srs_borrowers.fetch_sub(1) == 1 && lender->zombie) orphan = lender;
        c->srs_lender = nullptr;
        c->srs = nullptr; c->srs_n = 0; c->srs_w.release(); c->srs_w_valid = false; c->srs_w_copies = 0;
        if (c->lag_borrowed) { c->lag.pts = nullptr; c->lag.n = 0; c->lag.w.release(); c->lag.w_valid = false; c->lag.w_copies = 0; }
        c->lag_borrowed = false;
    }
    if (orphan) free_lent(orphan);                       // the lender was destroyed before this, its last, borrower
}
void srs_make_loan(plk_ctx *dst, plk_ctx *src) {         // (msm.hip plk_ctx_share_srs: the checks and the fallible steps come first)
    std::lock_guard<std::mutex> g(g_share_mu);
    dst->srs = src->srs; dst->srs_n = src->srs_n;
    dst->srs_w.borrow(src->srs_w); dst->srs_w_valid = src->srs_w_valid; dst->srs_w_copies = src->srs_w_copies;
    dst->lag.pts = src->lag.pts; dst->lag.n = src->lag.n;
    dst->lag.w.borrow(src->lag.w); dst->lag.w_valid = src->lag.pts ? src->lag.w_valid : false; dst->lag.w_copies = src->lag.w_copies;
    dst->lag_borrowed = src->lag.pts != nullptr;
    dst->srs_lender = src;
    src->srs_borrowers.fetch_add(1);
    if (dst->lag_borrowed) src->lag_borrowers.fetch_add(1);
}
// true: the context lends its key to contexts that are still alive — plk_destroy keeps the key and the tables (and the shell of the
// context) until the last of them returns its loan
bool srs_orphan_lender(plk_ctx *c) {
    std::lock_guard<std::mutex> g(g_share_mu);
    if (c->srs_borrowers.load() == 0) return false;
    c->zombie = true;
    return true;
}

}  // namespace plk

using namespace plk;

extern "C" {

const char *plk_last_error(void) { return g_last_error.c_str(); }
const char *plk_version(void) { return "plonkit_amd 0.1 (gfx950)"; }

// read by the runtime when it initialises, i.e. at the first HIP call of the process: more hardware queues than HIP's default of 4
// (a proof keeps 4-5 streams busy, several proofs may be in flight).  Never overrides the caller's setting.
static void runtime_env_defaults() {
    // once per process and only what a single-GPU user needs: the dmabuf IPC mode the multi-rank path needs
    // (HSA_ENABLE_IPC_MODE_LEGACY=0) changes the behaviour of every other HIP / RCCL user of the embedding process, so it is the
    // launcher's to set before the first HIP call — this package's binary (PLONKIT_WORLD > 1), bench.py and plonkit_amd.sharded do;
    // INTEGRATION.md names it.  (call_once: contexts are created from several host threads, setenv is not thread-safe.)
    static std::once_flag once;
    std::call_once(once, [] { setenv("GPU_MAX_HW_QUEUES", "8", 0); });
}

int32_t plk_device_count(void) {
    runtime_env_defaults();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int32_t plk_create(int32_t device, plk_ctx **out) {
    // up to three commitments may be in flight on three streams (MsmSlot): ask for enough hardware queues that they do not
    // share one (HIP's default is 4 per device; no effect if the runtime is already initialised by the host program)
    runtime_env_defaults();
    if (!out) { set_error("plk_create: null out"); return PLK_ERR_ARG; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        (void)hipGetLastError();
        set_error("plk_create: no HIP device visible — plonkit_amd has no CPU fallback (needs gfx950 / MI355X)");
        return PLK_ERR_HIP;
    }
    if (device < 0 || device >= n) { set_error("plk_create: device index out of range"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PLK_HIP(hipGetDeviceProperties(&prop, device));
    plk_ctx *ctx = new plk_ctx();
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return hip_fail(hipGetLastError(), "hipStreamCreate", __FILE__, __LINE__);
    }
    int32_t rc = ntt_init_tables(ctx);
    if (rc == PLK_OK) rc = ensure_pinned(ctx, 1 << 16);
    if (rc != PLK_OK) { plk_destroy(ctx); return rc; }
    *out = ctx;
    return PLK_OK;
}

void plk_destroy(plk_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); }
    for (auto &S : ctx->slot) if (S.stream) (void)hipStreamSynchronize(S.stream);     // commitments still in flight
    if (ctx->bg_stream) (void)hipStreamSynchronize(ctx->bg_stream);
    comm_release(ctx);
    srs_return_loan(ctx);
    const bool lent = srs_orphan_lender(ctx);            // borrowers alive: their commitments read this context's key and tables at any time
    for (void *p : ctx->coset_allocs) (void)hipFree(p);
    for (auto &kv : ctx->ntt_direct) (void)hipFree(kv.second);
    ctx->coset_direct[0].buf.release(); ctx->coset_direct[1].buf.release();
    ctx->tables.release(); ctx->ntt_scratch[0].release(); ctx->ntt_scratch[1].release();
    if (!lent) { ctx->srs_own.release(); ctx->srs_w.release(); ctx->lag.own.release(); ctx->lag.w.release(); }
    for (auto &S : ctx->slot) {
        S.a.release(); S.b.release(); S.c.release(); S.d.release(); S.e.release(); S.f.release();
        if (S.pinned) (void)hipHostFree(S.pinned);
        if (S.stream) (void)hipStreamDestroy(S.stream);
        if (S.ready) (void)hipEventDestroy(S.ready);
        if (S.acc_done) (void)hipEventDestroy(S.acc_done);
        if (S.ev[0]) { (void)hipEventDestroy(S.ev[0]); (void)hipEventDestroy(S.ev[1]); }
    }
    ctx->stage.release(); ctx->poly_tmp.release(); ctx->poly_tmp2.release(); ctx->prove_ws.release();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->pinned2) (void)hipHostFree(ctx->pinned2);
    if (ctx->flag_ready) (void)hipEventDestroy(ctx->flag_ready);
    if (ctx->bg_go) (void)hipEventDestroy(ctx->bg_go);
    if (ctx->bg_done) (void)hipEventDestroy(ctx->bg_done);
    if (ctx->bg_stream) (void)hipStreamDestroy(ctx->bg_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = nullptr; ctx->bg_stream = nullptr;
    if (!lent) delete ctx;                               // else: srs_return_loan of the last borrower frees the key, the tables and the shell
}

int32_t plk_synchronize(plk_ctx *ctx) {
    if (!ctx) { set_error("null ctx"); return PLK_ERR_ARG; }
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &S : ctx->slot) if (S.stream) PLK_HIP(hipStreamSynchronize(S.stream));
    return PLK_OK;
}

// ------------------------------------------------------------------------------------ SRS
int32_t plk_srs_upload(plk_ctx *ctx, const plk_g1_affine *bases, uint64_t n) {
    if (!ctx || !bases || n == 0) { set_error("plk_srs_upload: bad argument"); return PLK_ERR_ARG; }
    PLK_TRY(srs_replace_guard(ctx, "plk_srs_upload"));
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(ctx->srs_own.reserve(n * sizeof(plk_g1_affine)));
    PLK_HIP(hipMemcpyAsync(ctx->srs_own.p, bases, n * sizeof(plk_g1_affine), hipMemcpyHostToDevice, ctx->stream));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->srs = ctx->srs_own.p;
    ctx->srs_n = n;
    srs_table_invalidate(ctx);
    return PLK_OK;
}

int32_t plk_srs_set_dev(plk_ctx *ctx, const void *bases_dev, uint64_t n) {
    if (!ctx || !bases_dev || n == 0) { set_error("plk_srs_set_dev: bad argument"); return PLK_ERR_ARG; }
    PLK_TRY(srs_replace_guard(ctx, "plk_srs_set_dev"));
    ctx->srs = bases_dev;
    ctx->srs_n = n;
    srs_table_invalidate(ctx);
    return PLK_OK;
}

uint64_t plk_srs_size(const plk_ctx *ctx) { return ctx ? ctx->srs_n : 0; }

int32_t plk_set_commit_shard(plk_ctx *ctx, uint64_t first_index, plk_combine_fn combine, void *user) {
    if (!ctx) { set_error("plk_set_commit_shard: bad argument"); return PLK_ERR_ARG; }
    ctx->shard_first = combine ? first_index : 0;
    ctx->combine = combine;
    ctx->combine_user = user;
    return PLK_OK;
}

// Lagrange-form key (Crs<E, CrsForLagrangeForm>): second resident SRS, see include/plonkit_amd.h
int32_t plk_srs_lagrange_upload(plk_ctx *ctx, const plk_g1_affine *bases, uint64_t n) {
    if (!ctx || !bases || n == 0) { set_error("plk_srs_lagrange_upload: bad argument"); return PLK_ERR_ARG; }
    PLK_TRY(srs_replace_guard(ctx, "plk_srs_lagrange_upload", true));
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(ctx->lag.own.reserve(n * sizeof(plk_g1_affine)));
    PLK_HIP(hipMemcpyAsync(ctx->lag.own.p, bases, n * sizeof(plk_g1_affine), hipMemcpyHostToDevice, ctx->stream));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->lag.pts = ctx->lag.own.p; ctx->lag.n = n; lag_table_invalidate(ctx);
    return PLK_OK;
}
int32_t plk_srs_lagrange_set_dev(plk_ctx *ctx, const void *bases_dev, uint64_t n) {
    if (!ctx || !bases_dev || n == 0) { set_error("plk_srs_lagrange_set_dev: bad argument"); return PLK_ERR_ARG; }
    PLK_TRY(srs_replace_guard(ctx, "plk_srs_lagrange_set_dev", true));
    ctx->lag.pts = bases_dev; ctx->lag.n = n; lag_table_invalidate(ctx);
    return PLK_OK;
}
int32_t plk_srs_lagrange_clear(plk_ctx *ctx) {
    if (!ctx) { set_error("plk_srs_lagrange_clear: bad argument"); return PLK_ERR_ARG; }
    if (!ctx->lag.pts) return PLK_OK;                                    // nothing resident: nothing to replace (also on a lender)
    PLK_TRY(srs_replace_guard(ctx, "plk_srs_lagrange_clear", true));
    ctx->lag.pts = nullptr; ctx->lag.n = 0; lag_table_invalidate(ctx);
    return PLK_OK;
}
uint64_t plk_srs_lagrange_size(const plk_ctx *ctx) { return ctx ? ctx->lag.n : 0; }

// ------------------------------------------------------------------------------------ NTT
int32_t plk_ntt_dev(plk_ctx *ctx, void *data_dev, uint32_t log_n, int32_t inverse, const plk_fr *coset, void *stream) {
    if (!ctx) { set_error("null ctx"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    Fr g;
    if (coset) memcpy(&g, coset, 32);
    return ntt_dev(ctx, (Fr *)data_dev, log_n, inverse != 0, coset ? &g : nullptr, s);
}

int32_t plk_ntt(plk_ctx *ctx, plk_fr *data, uint32_t log_n, int32_t inverse, const plk_fr *coset) {
    if (!ctx || !data) { set_error("plk_ntt: bad argument"); return PLK_ERR_ARG; }
    if (log_n > MAX_LOG_N) { set_error("ntt: log_n exceeds the 2-adicity of Fr (28)"); return PLK_ERR_SIZE; }
    PLK_HIP(hipSetDevice(ctx->device));
    size_t bytes = sizeof(plk_fr) << log_n;
    PLK_TRY(ctx->stage.reserve(bytes));
    PLK_HIP(hipMemcpyAsync(ctx->stage.p, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    PLK_TRY(plk_ntt_dev(ctx, ctx->stage.p, log_n, inverse, coset, nullptr));
    PLK_HIP(hipMemcpyAsync(data, ctx->stage.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return PLK_OK;
}

int32_t plk_lde4_dev(plk_ctx *ctx, const void *coeffs_dev, uint32_t log_n, void *out_4n_dev, void *stream) {
    if (!ctx || !coeffs_dev || !out_4n_dev) { set_error("plk_lde4_dev: bad argument"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    return lde4_dev(ctx, (const Fr *)coeffs_dev, log_n, (Fr *)out_4n_dev, s);
}

int32_t plk_lde4_coset_major_dev(plk_ctx *ctx, const void *const *coeffs_dev, uint32_t count, uint32_t log_n, void *const *out_4n_dev, void *stream) {
    if (!ctx || !coeffs_dev || !out_4n_dev || count == 0 || count > 16) { set_error("plk_lde4_coset_major_dev: bad argument"); return PLK_ERR_ARG; }
    for (uint32_t k = 0; k < count; k++) if (!coeffs_dev[k] || !out_4n_dev[k]) { set_error("plk_lde4_coset_major_dev: null vector"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    return lde4cm_batch_dev(ctx, reinterpret_cast<const Fr *const *>(coeffs_dev), count, log_n, reinterpret_cast<Fr *const *>(out_4n_dev),
                            stream ? (hipStream_t)stream : ctx->stream, 0);
}

int32_t plk_icoset4_coset_major_dev(plk_ctx *ctx, void *data_4n_dev, uint32_t log_n, void *stream) {
    if (!ctx || !data_4n_dev) { set_error("plk_icoset4_coset_major_dev: bad argument"); return PLK_ERR_ARG; }
    if (log_n + 2 > 28) { set_error("plk_icoset4_coset_major_dev: 4n exceeds 2^28"); return PLK_ERR_SIZE; }
    PLK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    PLK_TRY(icoset4cm_dev(ctx, (Fr *)data_4n_dev, log_n, st, 0));
    // constants of the combine step: i^-1 (i = omega_4 = omega_4n^n) and 7^(-n c) / 4, in the 2^261 domain of the field layer
    using host::HFr;
    const uint64_t n = 1ull << log_n;
    HFr w4; { Fr w = ntt_omega(2); memcpy(w4.l, w.l, 32); }
    const HFr gN_inv = HFr::from_u64(7).pow_u64(n).inv(), two5 = HFr::from_u64(32);
    HFr c = HFr::from_u64(4).inv();
    Fr s_w[4], iinv;
    { HFr t = w4.inv() * two5; memcpy(iinv.l, t.l, 32); }
    for (int k = 0; k < 4; k++) { HFr t = c * two5; memcpy(s_w[k].l, t.l, 32); c = c * gN_inv; }
    return icoset_combine((Fr *)data_4n_dev, (uint32_t)n, iinv, s_w, st);
}

int32_t plk_lde4(plk_ctx *ctx, const plk_fr *coeffs, uint32_t log_n, plk_fr *out_4n) {
    if (!ctx || !coeffs || !out_4n) { set_error("plk_lde4: bad argument"); return PLK_ERR_ARG; }
    if (log_n + 2 > MAX_LOG_N) { set_error("lde4: 4n exceeds 2^28"); return PLK_ERR_SIZE; }
    PLK_HIP(hipSetDevice(ctx->device));
    size_t bytes = sizeof(plk_fr) << log_n;
    PLK_TRY(ctx->stage.reserve(5 * bytes));
    char *d_in = (char *)ctx->stage.p, *d_out = d_in + bytes;
    PLK_HIP(hipMemcpyAsync(d_in, coeffs, bytes, hipMemcpyHostToDevice, ctx->stream));
    PLK_TRY(lde4_dev(ctx, (const Fr *)d_in, log_n, (Fr *)d_out, ctx->stream));
    PLK_HIP(hipMemcpyAsync(out_4n, d_out, 4 * bytes, hipMemcpyDeviceToHost, ctx->stream));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return PLK_OK;
}

}  // extern "C"
