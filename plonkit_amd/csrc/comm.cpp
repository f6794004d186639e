// Multi-GPU commitments below the C ABI: one process per GPU, the SRS split into contiguous slices, every KZG
// commitment of plk_prove / plk_setup_write_vk = sum over ranks of MSM(slice of the scalars, slice of the SRS).
//
// The reference has one process and bellman's `Worker` thread pool (src/plonk.rs:41,47,183); the exchange step here is
// the one SURVEY.md §8(e) describes: ONE all-gather of the 96-byte Jacobian partial sums per batch of commitments over
// RCCL/xGMI (EC addition is not an RCCL reduction op, so a literal ncclAllReduce is impossible), then world-1 host EC
// additions per commitment.  The payload is <= 8 x 96 B per rank: latency-bound, nothing to tune in bandwidth terms.
//
// RCCL is bound at run time (dlopen of librccl.so.1) so that the library — and the `plonkit` binary for single-GPU use —
// load on machines without it, and so that a host program that already carries RCCL (PyTorch) shares its copy.
// A second transport, a TCP hub on 127.0.0.1, exists for the one case RCCL refuses: several ranks on the SAME device
// (the single-GPU test tier).  It moves the same bytes through the same combiner.  TEST TIER ONLY, not for production:
// the hub accepts any local process that connects and names a free rank (no authentication, no encryption, loopback only).
#include "ctx.h"
#include "comm.h"
#include "msm.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <poll.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <mutex>
#include <vector>

namespace plk {
using namespace host;

namespace {

struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    // scatter mode only (optional: a librccl without them keeps the replicate mode)
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                       // optional: the watchdog's way out
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr; // optional
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;         // optional: how many ranks RCCL itself sees (bench line)
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {                                 // (contexts on several host threads may come here at once: the load is whole or not at all)
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (!r.so) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.so, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.so, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.so, "ncclAllGather"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.so, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.so, "ncclGetErrorString"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.so, "ncclCommAbort"));
        r.CommGetAsyncError = reinterpret_cast<decltype(r.CommGetAsyncError)>(dlsym(r.so, "ncclCommGetAsyncError"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(r.so, "ncclBroadcast"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.so, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.so, "ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.so, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.so, "ncclGroupEnd"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.so, "ncclCommCount"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) { dlclose(r.so); r.so = nullptr; }
    });
    return r.so ? &r : nullptr;
}

// The host driver of this pool (and of any amdgpu that only supports dmabuf IPC) needs HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment BEFORE
// the process' first HIP call, or RCCL fails between processes with `hipIpcGetMemHandle: invalid argument` — the CLI, bench.py and sharded.py set
// it; a C-ABI embedder has to.  The library cannot set it late, so it says so where the failure surfaces.
std::string ipc_hint() {
    const char *e = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
    if (e && !strcmp(e, "0")) return "";
    return " (HSA_ENABLE_IPC_MODE_LEGACY is not 0 in this process: on dmabuf-only drivers RCCL fails between processes with "
           "`hipIpcGetMemHandle: invalid argument` unless it is set before the first HIP call)";
}
int32_t rccl_fail(ncclResult_t e, const char *what) {
    Rccl *R = rccl();
    set_error(std::string("RCCL: ") + what + ": " + (R && R->GetErrorString ? R->GetErrorString(e) : "error") + ipc_hint());
    return PLK_ERR_HIP;
}

// a rank that died must not leave the others blocked for ever: every socket operation gives up after COMM_TIMEOUT_S, and
// the RCCL exchange is watched by a deadline of its own (PLK_COMM_TIMEOUT_MS, default COMM_TIMEOUT_S) — the reference
// panics and exits when a worker fails (src/bin/main.rs:335,371,399); here the call returns PLK_ERR_HIP
constexpr int COMM_TIMEOUT_S = 180;
long comm_timeout_ms() {
    const char *e = getenv("PLK_COMM_TIMEOUT_MS");
    if (e && *e) { char *end = nullptr; long v = strtol(e, &end, 10); if (end != e && v > 0) return v; }      // (0 or junk: the default —
    return COMM_TIMEOUT_S * 1000L;                                       //  a zero deadline would fail every exchange on its first poll)
}
// PLK_COMM_IDLE_TIMEOUT_MS (default 0 = none): how long a worker of owner-computes mode waits for the owner's NEXT batch.  Waiting for
// work is not a fault — the owner may be between two proofs for hours — so there is no deadline unless the embedder sets one; a batch
// job (bench.py, the CLI under a launcher) sets it so that an owner that failed without reaching plk_comm_stop_workers cannot leave
// its workers parked in ncclBroadcast for ever (an aborted peer is NOT reported by ncclCommGetAsyncError inside one node).
long comm_idle_timeout_ms() {
    const char *e = getenv("PLK_COMM_IDLE_TIMEOUT_MS");
    if (e && *e) { char *end = nullptr; long v = strtol(e, &end, 10); if (end != e && v > 0) return v; }
    return 0;
}
// TEST HOOK (tests/test_gpu_sharded_prove.py): PLK_COMM_TEST_STALL_MS=<ms> parks the exchange stream for that long before
// every all-gather — a peer that does not answer —, so that the watchdog's abort path MUST run when the deadline is shorter
long comm_test_stall_ms() {
    static const long v = [] { const char *e = getenv("PLK_COMM_TEST_STALL_MS"); return e && *e ? strtol(e, nullptr, 10) : 0L; }();
    return v;
}
// (a kernel, not hipLaunchHostFunc: on this runtime a host function makes the enqueueing side wait for it)
__global__ void comm_test_stall_kernel(unsigned long long ticks) {           // wall_clock64: constant 100 MHz counter
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
// plk_comm_scatter_selftest: n scalars (8 words each, < 2^252) that depend on `salt`
__global__ void comm_test_fill_kernel(uint32_t *v, uint64_t n, uint32_t salt) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i * 0x9e3779b9u + salt * 0x85ebca6bu + 0x27d4eb2fu;
    for (int k = 0; k < 8; k++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v[8 * i + k] = k == 7 ? (x & 0x0fffffffu) : x; }
}
// PLK_SHARD_MODE=scatter: communicators start in owner-computes mode (every rank must agree); plk_comm_set_mode overrides
bool shard_mode_default() { const char *e = getenv("PLK_SHARD_MODE"); return e && !strcmp(e, "scatter"); }
void set_timeouts(int fd) {
    timeval tv{COMM_TIMEOUT_S, 0};
    ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    ::setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
}

bool send_all(int fd, const void *p, size_t n) {
    const char *c = static_cast<const char *>(p);
    while (n) { ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL); if (k <= 0) return false; c += k; n -= (size_t)k; }
    return true;
}
bool recv_all(int fd, void *p, size_t n) {
    char *c = static_cast<char *>(p);
    while (n) { ssize_t k = ::recv(fd, c, n, 0); if (k <= 0) return false; c += k; n -= (size_t)k; }
    return true;
}

}  // namespace

struct Comm {
    int rank = 0, world = 1;
    // RCCL transport
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;            // the exchange has a stream of its own: synchronising the context's main stream
    DevBuf d_send, d_recv;                   // would wait for whatever the prover overlaps with the commitments (LDEs)
    void *h_pin = nullptr; size_t h_pin_cap = 0;   // page-locked staging of both directions: a copy to or from PAGEABLE memory is synchronous —
                                             // the host would sit inside hipMemcpyAsync behind a collective that never ends, and the watchdog
                                             // below would never get to run (found with PLK_COMM_TEST_STALL_MS in round 4)
    // TCP hub transport (rank 0 listens; fds[r] = connection of rank r on the hub, fds[0] = the hub's socket on a spoke)
    bool tcp = false;
    int listen_fd = -1;
    std::vector<int> fds;
    std::vector<plk_g1_jacobian> host_all;
    plk_ctx *ctx = nullptr;
    uint64_t gathers = 0;                    // number of exchanges performed (tracing / tests)
    bool scatter = false;                    // owner-computes mode (PLK_SHARD_MODE=scatter / plk_comm_set_mode): see comm.h
    DevBuf d_hdr, d_work;                    // scatter mode: the 64-byte header, a worker's received slices
    hipEvent_t ev = nullptr;                 // orders the exchange stream against the producer / consumer stream of the scalars
    void *h_hdr = nullptr;                   // page-locked home of the batch header, apart from h_pin: the owner's header copy is asynchronous and may still be
                                             // pending (its stream waits for the prover's) when the batch's gather stages its partial sums in h_pin
    std::vector<char> h_stage;               // TCP transport of the slices (test tier)
    bool self_loop = false;                  // plk_comm_scatter_selftest: rank 0 is also the recipient of its own share (the RCCL branch on one GPU)
    uint64_t seq = 0;                        // batches sent / received (both sides count: a header out of step is a protocol error)
    bool dead = false;                       // the watchdog aborted the communicator: every later exchange fails at once
    bool leak = false;                       // ... and could not abort the collective: stream and buffers are abandoned, not freed
};

// Waits for the exchange stream with a deadline instead of hipStreamSynchronize: if a peer died, ncclAllGather never
// completes and a plain synchronise would hang this rank (and with it the node) for ever.  Spins for the common case
// (the exchange takes tens of microseconds), then yields, then sleeps.  On expiry or an asynchronous RCCL error the
// communicator is aborted (ncclCommAbort tears the collective's kernel down so that the stream drains) and marked dead.
static int32_t watch_exchange(Comm *C, hipStream_t st, bool no_deadline = false) {
    using clk = std::chrono::steady_clock;
    Rccl *R = rccl();
    const auto t0 = clk::now();
    // (no_deadline: a worker waiting for the owner's next batch — the owner may be inside its transforms, or between two proofs, for
    //  any length of time: only PLK_COMM_IDLE_TIMEOUT_MS, if set, bounds it; a peer PROCESS that died may end the wait through
    //  ncclCommGetAsyncError)
    const long idle_ms = no_deadline ? comm_idle_timeout_ms() : 0;
    const auto deadline = !no_deadline ? t0 + std::chrono::milliseconds(comm_timeout_ms())
                                       : (idle_ms > 0 ? t0 + std::chrono::milliseconds(idle_ms) : clk::time_point::max());
    for (unsigned it = 0;; it++) {
        hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return PLK_OK;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); C->dead = true; set_error(std::string("RCCL exchange: ") + hipGetErrorString(q)); return PLK_ERR_HIP; }
        (void)hipGetLastError();
        const auto now = clk::now();
        bool failed = now >= deadline;
        const char *why = no_deadline ? "no batch from the owner before the idle deadline (PLK_COMM_IDLE_TIMEOUT_MS)"
                                      : "no answer from a peer before the deadline (PLK_COMM_TIMEOUT_MS)";
        if (!failed && (it & 1023) == 1023 && R->CommGetAsyncError) {
            ncclResult_t ae = ncclSuccess;
            if (R->CommGetAsyncError(C->nccl, &ae) == ncclSuccess && ae != ncclSuccess && ae != ncclInProgress) { failed = true; why = "asynchronous RCCL error (a peer went away)"; }
        }
        if (failed) {
            C->dead = true;
            if (R->CommAbort && C->nccl) { (void)R->CommAbort(C->nccl); C->nccl = nullptr; }
            else C->leak = true;              // no ncclCommAbort in this librccl: the collective may never drain — comm_free must not
                                              // wait for it (hipStreamDestroy / hipFree would hang the way the watchdog exists to avoid)
            set_error(std::string("RCCL exchange aborted: ") + why);
            return PLK_ERR_HIP;
        }
        const auto waited = now - t0;
        if (waited > std::chrono::milliseconds(20)) std::this_thread::sleep_for(std::chrono::microseconds(200));
        else if (waited > std::chrono::microseconds(300)) std::this_thread::yield();
    }
}


// all ranks' `count` partial sums -> all[r * count + k]
static int32_t gather(Comm *C, const plk_g1_jacobian *mine, uint32_t count, plk_g1_jacobian *all) {
    const size_t bytes = (size_t)count * sizeof(plk_g1_jacobian);
    C->gathers++;
    if (C->tcp) {
        if (C->rank == 0) {
            memcpy(all, mine, bytes);
            for (int r = 1; r < C->world; r++)
                if (!recv_all(C->fds[r], reinterpret_cast<char *>(all) + (size_t)r * bytes, bytes)) { set_error("tcp combiner: a rank went away"); return PLK_ERR_IO; }
            for (int r = 1; r < C->world; r++)
                if (!send_all(C->fds[r], all, bytes * C->world)) { set_error("tcp combiner: a rank went away"); return PLK_ERR_IO; }
        } else {
            if (!send_all(C->fds[0], mine, bytes) || !recv_all(C->fds[0], all, bytes * C->world)) { set_error("tcp combiner: the hub went away"); return PLK_ERR_IO; }
        }
        return PLK_OK;
    }
    Rccl *R = rccl();
    if (C->dead || !C->nccl) { set_error("RCCL exchange: the communicator was aborted earlier (plk_comm_destroy + plk_comm_init to start over)"); return PLK_ERR_HIP; }
    PLK_HIP(hipSetDevice(C->ctx->device));
    hipStream_t st = C->stream;
    PLK_TRY(C->d_send.reserve(8 * sizeof(plk_g1_jacobian)));
    PLK_TRY(C->d_recv.reserve((size_t)C->world * 8 * sizeof(plk_g1_jacobian)));
    const size_t pin_need = bytes * ((size_t)C->world + 1);
    if (C->h_pin_cap < pin_need) {
        if (C->h_pin) (void)hipHostFree(C->h_pin);
        C->h_pin = nullptr; C->h_pin_cap = 0;
        PLK_HIP(hipHostMalloc(&C->h_pin, (size_t)8 * sizeof(plk_g1_jacobian) * ((size_t)C->world + 1), hipHostMallocDefault));
        C->h_pin_cap = (size_t)8 * sizeof(plk_g1_jacobian) * ((size_t)C->world + 1);
    }
    char *pin_send = static_cast<char *>(C->h_pin), *pin_recv = pin_send + bytes;
    memcpy(pin_send, mine, bytes);
    PLK_HIP(hipMemcpyAsync(C->d_send.p, pin_send, bytes, hipMemcpyHostToDevice, st));
    if (comm_test_stall_ms() > 0) hipLaunchKernelGGL(comm_test_stall_kernel, dim3(1), dim3(1), 0, st, (unsigned long long)comm_test_stall_ms() * 100000ull);
    ncclResult_t e = R->AllGather(C->d_send.p, C->d_recv.p, bytes, ncclUint8, C->nccl, st);
    if (e != ncclSuccess) return rccl_fail(e, "ncclAllGather");
    PLK_HIP(hipMemcpyAsync(pin_recv, C->d_recv.p, bytes * C->world, hipMemcpyDeviceToHost, st));
    PLK_TRY(watch_exchange(C, st));
    memcpy(all, pin_recv, bytes * C->world);
    return PLK_OK;
}

// the built-in plk_combine_fn: sums[k] <- sum over ranks of sums[k]; every rank ends with the same group elements
static int32_t builtin_combine(void *user, plk_g1_jacobian *sums, uint32_t count) {
    Comm *C = static_cast<Comm *>(user);
    if (count == 0 || count > 8) { set_error("combiner: batch must be 1..8"); return PLK_ERR_ARG; }
    C->host_all.resize((size_t)C->world * count);
    PLK_TRY(gather(C, sums, count, C->host_all.data()));
    for (uint32_t k = 0; k < count; k++) {
        HJac acc = HJac::inf();
        for (int r = 0; r < C->world; r++) {
            const plk_g1_jacobian &p = C->host_all[(size_t)r * count + k];
            HJac j; memcpy(j.x.l, p.x, 32); memcpy(j.y.l, p.y, 32); memcpy(j.z.l, p.z, 32);
            acc = jac_add(acc, j);                     // rank order 0..world-1 on every rank: identical bytes everywhere
        }
        memcpy(sums[k].x, acc.x.l, 32); memcpy(sums[k].y, acc.y.l, 32); memcpy(sums[k].z, acc.z.l, 32);
    }
    return PLK_OK;
}

static void comm_free(Comm *C) {
    if (!C) return;
    if (C->leak) { C->stream = nullptr; C->d_send.p = nullptr; C->d_recv.p = nullptr; C->d_hdr.p = nullptr; C->d_work.p = nullptr; C->nccl = nullptr; C->h_pin = nullptr; C->h_hdr = nullptr; }      // see watch_exchange
    if (C->nccl) { Rccl *R = rccl(); if (R) (void)R->CommDestroy(C->nccl); }
    if (C->stream) (void)hipStreamDestroy(C->stream);
    C->d_send.release(); C->d_recv.release(); C->d_hdr.release(); C->d_work.release();
    if (C->ev) (void)hipEventDestroy(C->ev);
    if (C->h_pin) (void)hipHostFree(C->h_pin);
    if (C->h_hdr) (void)hipHostFree(C->h_hdr);
    for (int fd : C->fds) if (fd >= 0) ::close(fd);
    if (C->listen_fd >= 0) ::close(C->listen_fd);
    delete C;
}

// ------------------------------------------------------------------------------- owner-computes mode: the transport
// One batch = a 64-byte header from rank 0 to everyone, then rank 0 -> rank r: count x (r's share of the vector) x 32 bytes, grouped
// ncclSend / ncclRecv on the exchange stream (xGMI is point to point: the seven slices leave on seven links at once — 4 MiB per link
// and vector at the 2^20 domain, 64 MiB at 2^24), then the all-gather of the 96-byte partial sums every mode ends a batch with.
struct ShardHeader { uint32_t magic, op, count, lagrange; uint64_t n, slice, seq; uint8_t pad[24]; };
static_assert(sizeof(ShardHeader) == 64, "64-byte header");
constexpr uint32_t SHARD_MAGIC = 0x706c6b53u;          // "plkS"

static bool scatter_rank(const plk_ctx *ctx, bool owner) {
    const Comm *C = ctx ? static_cast<const Comm *>(ctx->comm) : nullptr;
    return C && C->scatter && C->world > 1 && (owner ? C->rank == 0 : C->rank > 0);
}
bool comm_scatter_owner(const plk_ctx *ctx) { return scatter_rank(ctx, true); }
bool comm_scatter_worker(const plk_ctx *ctx) { return scatter_rank(ctx, false); }

static int32_t scatter_ready(Comm *C) {
    if (C->tcp) return PLK_OK;
    Rccl *R = rccl();
    if (!R || !R->Broadcast || !R->Send || !R->Recv || !R->GroupStart || !R->GroupEnd) { set_error("scatter mode: this librccl lacks ncclBroadcast / ncclSend / ncclRecv / ncclGroupStart"); return PLK_ERR_HIP; }
    if (C->dead || !C->nccl) { set_error("RCCL exchange: the communicator was aborted earlier (plk_comm_destroy + plk_comm_init to start over)"); return PLK_ERR_HIP; }
    PLK_HIP(hipSetDevice(C->ctx->device));
    PLK_TRY(C->d_hdr.reserve(sizeof(ShardHeader)));
    if (!C->ev) PLK_HIP(hipEventCreateWithFlags(&C->ev, hipEventDisableTiming));
    if (!C->h_hdr) PLK_HIP(hipHostMalloc(&C->h_hdr, 2 * sizeof(ShardHeader), hipHostMallocDefault));
    return PLK_OK;
}

// rank 0 -> everyone
static int32_t send_header(Comm *C, const ShardHeader &h) {
    if (C->tcp) {
        for (int r = 1; r < C->world; r++) if (!send_all(C->fds[r], &h, sizeof h)) { set_error("tcp scatter: a rank went away"); return PLK_ERR_IO; }
        return PLK_OK;
    }
    Rccl *R = rccl();
    memcpy(C->h_hdr, &h, sizeof h);
    PLK_HIP(hipMemcpyAsync(C->d_hdr.p, C->h_hdr, sizeof h, hipMemcpyHostToDevice, C->stream));
    ncclResult_t e = R->Broadcast(C->d_hdr.p, C->d_hdr.p, sizeof h, ncclUint8, 0, C->nccl, C->stream);
    if (e != ncclSuccess) return rccl_fail(e, "ncclBroadcast (batch header)");
    return PLK_OK;
}
static int32_t recv_header(Comm *C, ShardHeader *h) {
    if (C->tcp) {
        const long idle_ms = comm_idle_timeout_ms();                 // (waiting for work is not a fault: no receive timeout unless one is set)
        timeval none{idle_ms / 1000, (idle_ms % 1000) * 1000};
        ::setsockopt(C->fds[0], SOL_SOCKET, SO_RCVTIMEO, &none, sizeof none);
        const bool ok = recv_all(C->fds[0], h, sizeof *h);
        set_timeouts(C->fds[0]);
        if (!ok) { set_error("tcp scatter: the owner went away"); return PLK_ERR_IO; }
        return PLK_OK;
    }
    Rccl *R = rccl();
    ncclResult_t e = R->Broadcast(C->d_hdr.p, C->d_hdr.p, sizeof *h, ncclUint8, 0, C->nccl, C->stream);
    if (e != ncclSuccess) return rccl_fail(e, "ncclBroadcast (batch header)");
    PLK_HIP(hipMemcpyAsync(C->h_hdr, C->d_hdr.p, sizeof *h, hipMemcpyDeviceToHost, C->stream));
    PLK_TRY(watch_exchange(C, C->stream, true));
    memcpy(h, C->h_hdr, sizeof *h);
    return PLK_OK;
}
static uint64_t share_of(uint64_t n, uint64_t slice, int rank) {
    const uint64_t lo = (uint64_t)rank * slice;
    return lo >= n ? 0 : (n - lo < slice ? n - lo : slice);
}

int32_t comm_send_work(plk_ctx *ctx, const void *const *vecs, uint32_t count, uint64_t n, uint64_t slice, bool lagrange, hipStream_t producer) {
    Comm *C = static_cast<Comm *>(ctx->comm);
    if (count == 0 || count > 8) { set_error("scatter: batch must be 1..8"); return PLK_ERR_ARG; }
    if (slice == 0) { set_error("scatter: no key resident on the owner"); return PLK_ERR_SRS; }
    PLK_TRY(scatter_ready(C));
    // (arguments are checked before anything is counted or posted: a call that fails without reaching the wire must leave the owner's
    //  sequence number where the workers' is — the next valid batch would otherwise be refused as "header out of step")
    ShardHeader h{};
    h.magic = SHARD_MAGIC; h.op = SHARD_COMMIT; h.count = count; h.lagrange = lagrange ? 1 : 0;
    h.n = n; h.slice = slice; h.seq = C->seq + 1;
    if (C->tcp) {                                                    // test tier: through host memory
        PLK_HIP(hipStreamSynchronize(producer));
        PLK_TRY(send_header(C, h));
        C->seq++;
        for (int r = 1; r < C->world; r++) {
            const uint64_t len = share_of(n, h.slice, r);
            if (!len) continue;
            C->h_stage.resize((size_t)len * 32);
            for (uint32_t k = 0; k < count; k++) {
                PLK_HIP(hipMemcpy(C->h_stage.data(), static_cast<const char *>(vecs[k]) + (size_t)r * h.slice * 32, (size_t)len * 32, hipMemcpyDeviceToHost));
                if (!send_all(C->fds[r], C->h_stage.data(), (size_t)len * 32)) { C->dead = true; set_error("tcp scatter: a rank went away"); return PLK_ERR_IO; }
            }
        }
        return PLK_OK;
    }
    Rccl *R = rccl();
    // the scalars are still being written on the prover's stream: everything posted below waits for an event recorded behind them
    // (PLK_COMM_TEST_SKIP_PRODUCER_WAIT=1, TEST HOOK of plk_comm_scatter_selftest's negative control, leaves the wait out)
    static const bool skip_wait = [] { const char *e = getenv("PLK_COMM_TEST_SKIP_PRODUCER_WAIT"); return e && e[0] == '1'; }();
    PLK_HIP(hipEventRecord(C->ev, producer));
    if (!skip_wait) PLK_HIP(hipStreamWaitEvent(C->stream, C->ev, 0));
    PLK_TRY(send_header(C, h));
    C->seq++;                                                        // the header is on the wire: from here on a failure kills the communicator
    uint64_t self_len = 0;
    if (C->self_loop) {                                              // one GPU: rank 0 receives its own share (send and receive in ONE group)
        self_len = share_of(n, h.slice, 0);
        if (C->d_work.reserve((size_t)count * self_len * 32) != PLK_OK) { C->dead = true; return PLK_ERR_HIP; }
    }
    ncclResult_t e = R->GroupStart();
    if (e != ncclSuccess) { C->dead = true; return rccl_fail(e, "ncclGroupStart"); }
    for (int r = C->self_loop ? 0 : 1; r < (C->self_loop ? 1 : C->world) && e == ncclSuccess; r++) {
        const uint64_t len = share_of(n, h.slice, r);
        for (uint32_t k = 0; k < count && len && e == ncclSuccess; k++) {
            e = R->Send(static_cast<const char *>(vecs[k]) + (size_t)r * h.slice * 32, (size_t)len * 32, ncclUint8, r, C->nccl, C->stream);
            if (C->self_loop && e == ncclSuccess) e = R->Recv(static_cast<char *>(C->d_work.p) + (size_t)k * len * 32, (size_t)len * 32, ncclUint8, 0, C->nccl, C->stream);
        }
    }
    const ncclResult_t e2 = R->GroupEnd();
    if (e != ncclSuccess) { C->dead = true; return rccl_fail(e, "ncclSend (scalar slices)"); }
    if (e2 != ncclSuccess) { C->dead = true; return rccl_fail(e2, "ncclGroupEnd"); }
    return PLK_OK;                                                   // (not waited for: the batch's all-gather follows on the same stream)
}

int32_t comm_recv_work(plk_ctx *ctx, ShardWork *w, hipStream_t consumer) {
    Comm *C = static_cast<Comm *>(ctx->comm);
    PLK_TRY(scatter_ready(C));
    ShardHeader h{};
    PLK_TRY(recv_header(C, &h));
    // (self loop: the same process already counted this batch when it sent it)
    if (h.magic != SHARD_MAGIC || (h.op != SHARD_COMMIT && h.op != SHARD_STOP) || h.seq != (C->self_loop ? C->seq : ++C->seq)) { set_error("scatter: batch header out of step (do all ranks run the same mode?)"); return PLK_ERR_IO; }
    *w = ShardWork();
    w->op = h.op;
    if (h.op == SHARD_STOP) return PLK_OK;
    if (h.count == 0 || h.count > 8 || h.slice == 0) { set_error("scatter: malformed batch header"); return PLK_ERR_IO; }
    w->count = h.count; w->lagrange = h.lagrange; w->n = h.n; w->slice = h.slice;
    w->len = share_of(h.n, h.slice, C->rank);
    if (!w->len) return PLK_OK;
    PLK_TRY(C->d_work.reserve((size_t)h.count * w->len * 32));
    for (uint32_t k = 0; k < h.count; k++) w->vec[k] = static_cast<char *>(C->d_work.p) + (size_t)k * w->len * 32;
    if (C->tcp) {
        C->h_stage.resize((size_t)w->len * 32);
        for (uint32_t k = 0; k < h.count; k++) {
            if (!recv_all(C->fds[0], C->h_stage.data(), (size_t)w->len * 32)) { set_error("tcp scatter: the owner went away"); return PLK_ERR_IO; }
            PLK_HIP(hipMemcpy(const_cast<void *>(w->vec[k]), C->h_stage.data(), (size_t)w->len * 32, hipMemcpyHostToDevice));
        }
        return PLK_OK;
    }
    Rccl *R = rccl();
    if (!C->self_loop) {                                             // (self loop: the receives were posted in the sender's group)
        ncclResult_t e = R->GroupStart();
        if (e != ncclSuccess) return rccl_fail(e, "ncclGroupStart");
        for (uint32_t k = 0; k < h.count && e == ncclSuccess; k++) e = R->Recv(const_cast<void *>(w->vec[k]), (size_t)w->len * 32, ncclUint8, 0, C->nccl, C->stream);
        const ncclResult_t e2 = R->GroupEnd();
        if (e != ncclSuccess) return rccl_fail(e, "ncclRecv (scalar slices)");
        if (e2 != ncclSuccess) return rccl_fail(e2, "ncclGroupEnd");
    }
    PLK_HIP(hipEventRecord(C->ev, C->stream));                       // the commitment's kernels read the slices on the consumer's stream
    PLK_HIP(hipStreamWaitEvent(consumer, C->ev, 0));
    return PLK_OK;
}

int32_t comm_send_stop(plk_ctx *ctx) {
    Comm *C = static_cast<Comm *>(ctx->comm);
    PLK_TRY(scatter_ready(C));
    ShardHeader h{};
    h.magic = SHARD_MAGIC; h.op = SHARD_STOP; h.seq = C->seq + 1;
    PLK_TRY(send_header(C, h));
    C->seq++;
    if (!C->tcp) PLK_TRY(watch_exchange(C, C->stream));
    return PLK_OK;
}

void comm_release(plk_ctx *ctx) {
    if (!ctx || !ctx->comm) return;
    if (ctx->combine == builtin_combine) { ctx->combine = nullptr; ctx->combine_user = nullptr; ctx->shard_first = 0; }
    comm_free(static_cast<Comm *>(ctx->comm));
    ctx->comm = nullptr;
}

}  // namespace plk

using namespace plk;

extern "C" {

int32_t plk_comm_unique_id(plk_comm_id *out) {
    if (!out) { set_error("plk_comm_unique_id: null"); return PLK_ERR_ARG; }
    static_assert(sizeof(plk_comm_id) == sizeof(ncclUniqueId), "plk_comm_id is an ncclUniqueId");
    Rccl *R = rccl();
    if (!R) { set_error("RCCL (librccl.so.1) cannot be loaded"); return PLK_ERR_HIP; }
    ncclUniqueId id;
    ncclResult_t e = R->GetUniqueId(&id);
    if (e != ncclSuccess) return rccl_fail(e, "ncclGetUniqueId");
    memcpy(out->bytes, id.internal, sizeof id.internal);
    return PLK_OK;
}

int32_t plk_comm_init(plk_ctx *ctx, int32_t rank, int32_t world, const plk_comm_id *id, uint64_t first_index) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) { set_error("plk_comm_init: bad argument"); return PLK_ERR_ARG; }
    Rccl *R = rccl();
    if (!R) { set_error("RCCL (librccl.so.1) cannot be loaded"); return PLK_ERR_HIP; }
    PLK_HIP(hipSetDevice(ctx->device));
    comm_release(ctx);
    Comm *C = new Comm();
    C->rank = rank; C->world = world; C->ctx = ctx;
    ncclUniqueId nid;
    memcpy(nid.internal, id->bytes, sizeof nid.internal);
    ncclResult_t e = R->CommInitRank(&C->nccl, world, nid, rank);
    if (e != ncclSuccess) { C->nccl = nullptr; comm_free(C); return rccl_fail(e, "ncclCommInitRank (one rank per GPU: RCCL refuses two ranks on one device)"); }
    if (hipStreamCreateWithFlags(&C->stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); comm_free(C); set_error("plk_comm_init: cannot create a stream"); return PLK_ERR_HIP; }
    ctx->comm = C;
    C->scatter = shard_mode_default();
    return plk_set_commit_shard(ctx, first_index, builtin_combine, C);
}

int32_t plk_comm_open_tcp(int32_t rank, int32_t world, uint16_t port, void **out) {
    if (!out || world < 1 || rank < 0 || rank >= world || port == 0) { set_error("plk_comm_open_tcp: bad argument"); return PLK_ERR_ARG; }
    *out = nullptr;
    Comm *C = new Comm();
    C->rank = rank; C->world = world; C->ctx = nullptr; C->tcp = true;
    C->fds.assign(world, -1);
    sockaddr_in addr{};
    addr.sin_family = AF_INET; addr.sin_port = htons(port); addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    const int one = 1;
    if (rank == 0) {
        C->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
        ::setsockopt(C->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        if (C->listen_fd < 0 || ::bind(C->listen_fd, reinterpret_cast<sockaddr *>(&addr), sizeof addr) != 0 || ::listen(C->listen_fd, world) != 0) {
            comm_free(C); set_error("plk_comm_open_tcp: cannot listen on 127.0.0.1"); return PLK_ERR_IO; }
        for (int k = 1; k < world; k++) {
            pollfd pf{C->listen_fd, POLLIN, 0};
            int fd = ::poll(&pf, 1, COMM_TIMEOUT_S * 1000) > 0 ? ::accept(C->listen_fd, nullptr, nullptr) : -1;
            if (fd >= 0) set_timeouts(fd);
            int32_t peer = -1;
            if (fd < 0 || !recv_all(fd, &peer, 4) || peer < 1 || peer >= world || C->fds[peer] >= 0) { if (fd >= 0) ::close(fd); comm_free(C); set_error("plk_comm_init_tcp: bad peer"); return PLK_ERR_IO; }
            ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
            C->fds[peer] = fd;
        }
    } else {
        int fd = -1;
        for (int attempt = 0; attempt < 600; attempt++) {            // the hub may start later: retry for a minute
            fd = ::socket(AF_INET, SOCK_STREAM, 0);
            if (fd >= 0 && ::connect(fd, reinterpret_cast<sockaddr *>(&addr), sizeof addr) == 0) break;
            if (fd >= 0) ::close(fd);
            fd = -1;
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        int32_t me = rank;
        if (fd >= 0) set_timeouts(fd);
        if (fd < 0 || !send_all(fd, &me, 4)) { if (fd >= 0) ::close(fd); comm_free(C); set_error("plk_comm_init_tcp: cannot reach rank 0"); return PLK_ERR_IO; }
        ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        C->fds[0] = fd;
    }
    *out = C;
    return PLK_OK;
}

void plk_comm_close(void *comm) { comm_free(static_cast<Comm *>(comm)); }

// the scatter step of owner-computes mode on HOST buffers over a TCP communicator (plk_comm_open_tcp): same header, same shares, same
// sequence numbers as comm_send_work / comm_recv_work, without the device copies — what the CPU tests drive (tests/test_sharded_gloo.py).
// rank 0: vecs[0..count) of n elements (32 B each) -> every rank r > 0 receives [r * slice, min((r + 1) * slice, n)) of each; count = 0 sends
// the stop message.  rank > 0: blocks for the next message; *count_out = 0 means stop, else count_out x len_out x 32 bytes are in `mine`.
int32_t plk_comm_scatter_host(void *comm, const void *const *vecs, uint32_t count, uint64_t n, uint64_t slice,
                              void *mine, uint64_t mine_cap, uint32_t *count_out, uint64_t *len_out) {
    Comm *C = static_cast<Comm *>(comm);
    if (!C || !C->tcp || count > 8) { set_error("plk_comm_scatter_host: needs a TCP communicator (plk_comm_open_tcp) and at most 8 vectors"); return PLK_ERR_ARG; }
    if (C->rank == 0) {
        if (count && (!vecs || slice == 0)) { set_error("plk_comm_scatter_host: bad argument"); return PLK_ERR_ARG; }
        ShardHeader h{};
        h.magic = SHARD_MAGIC; h.op = count ? SHARD_COMMIT : SHARD_STOP; h.count = count; h.n = n; h.slice = slice; h.seq = C->seq + 1;
        PLK_TRY(send_header(C, h));
        C->seq++;
        for (int r = 1; r < C->world && count; r++) {
            const uint64_t len = share_of(n, slice, r);
            for (uint32_t k = 0; k < count && len; k++)
                if (!send_all(C->fds[r], static_cast<const char *>(vecs[k]) + (size_t)r * slice * 32, (size_t)len * 32)) { set_error("tcp scatter: a rank went away"); return PLK_ERR_IO; }
        }
        if (count_out) *count_out = count;
        if (len_out) *len_out = share_of(n, slice, 0);
        return PLK_OK;
    }
    if (!count_out || !len_out) { set_error("plk_comm_scatter_host: bad argument"); return PLK_ERR_ARG; }
    ShardHeader h{};
    PLK_TRY(recv_header(C, &h));
    if (h.magic != SHARD_MAGIC || (h.op != SHARD_COMMIT && h.op != SHARD_STOP) || h.seq != ++C->seq) { set_error("scatter: batch header out of step"); return PLK_ERR_IO; }
    *count_out = h.op == SHARD_STOP ? 0 : h.count;
    *len_out = h.op == SHARD_STOP ? 0 : share_of(h.n, h.slice, C->rank);
    if ((uint64_t)*count_out * *len_out * 32 > mine_cap || (*count_out && *len_out && !mine)) { set_error("plk_comm_scatter_host: receive buffer too small"); return PLK_ERR_ARG; }
    for (uint32_t k = 0; k < *count_out && *len_out; k++)
        if (!recv_all(C->fds[0], static_cast<char *>(mine) + (size_t)k * *len_out * 32, (size_t)*len_out * 32)) { set_error("tcp scatter: the owner went away"); return PLK_ERR_IO; }
    return PLK_OK;
}

int32_t plk_comm_combine(void *comm, plk_g1_jacobian *sums, uint32_t count) {
    if (!comm || !sums) { set_error("plk_comm_combine: bad argument"); return PLK_ERR_ARG; }
    return builtin_combine(comm, sums, count);
}

int32_t plk_comm_init_tcp(plk_ctx *ctx, int32_t rank, int32_t world, uint16_t port, uint64_t first_index) {
    if (!ctx) { set_error("plk_comm_init_tcp: bad argument"); return PLK_ERR_ARG; }
    comm_release(ctx);
    void *C = nullptr;
    PLK_TRY(plk_comm_open_tcp(rank, world, port, &C));
    static_cast<Comm *>(C)->ctx = ctx;
    static_cast<Comm *>(C)->scatter = shard_mode_default();
    ctx->comm = C;
    return plk_set_commit_shard(ctx, first_index, builtin_combine, C);
}

int32_t plk_comm_set_mode(plk_ctx *ctx, int32_t mode) {
    if (!ctx || !ctx->comm || (mode != PLK_SHARD_REPLICATE && mode != PLK_SHARD_SCATTER)) { set_error("plk_comm_set_mode: no communicator on this context, or an unknown mode"); return PLK_ERR_ARG; }
    Comm *C = static_cast<Comm *>(ctx->comm);
    if (mode == PLK_SHARD_SCATTER && !C->tcp && C->world > 1 && !C->scatter) {
        // EXPERIMENTAL between GPUs (no multi-GPU node has run it: DESIGN.md section 6): every point-to-point piece of the mode once — a collective,
        // like this call (every rank switches modes) — before the mode is trusted; a communicator that cannot do it stays in replicate mode
        const int32_t rc = plk_comm_selftest(ctx);
        if (rc != PLK_OK) { const std::string why = plk_last_error(); set_error("plk_comm_set_mode: owner-computes mode refused, its transport failed the self-test: " + why); return rc; }
    }
    C->scatter = mode == PLK_SHARD_SCATTER;
    return PLK_OK;
}

int32_t plk_comm_stop_workers(plk_ctx *ctx) {
    if (!comm_scatter_owner(ctx)) { set_error("plk_comm_stop_workers: not the owner (rank 0) of a communicator in scatter mode"); return PLK_ERR_ARG; }
    return comm_send_stop(ctx);
}

// Every point-to-point piece of owner-computes mode once, on any communicator (world >= 1, every rank calls it): a 64-byte header
// broadcast from rank 0, then a ring step — rank r sends 4 KiB to rank r + 1 and receives from rank r - 1 in ONE group (with one rank:
// to and from itself) — checked word by word.  What a deployment runs on a new node before it trusts the mode (bench.py does, before its
// owner-computes leg), and the only way the RCCL side of that transport can run on a single GPU.
int32_t plk_comm_selftest(plk_ctx *ctx) {
    Comm *C = ctx ? static_cast<Comm *>(ctx->comm) : nullptr;
    if (!C || C->tcp) { set_error("plk_comm_selftest: needs an RCCL communicator on this context (plk_comm_init)"); return PLK_ERR_ARG; }
    PLK_TRY(scatter_ready(C));
    Rccl *R = rccl();
    constexpr size_t WORDS = 1024;
    PLK_TRY(C->d_work.reserve(2 * WORDS * sizeof(uint32_t)));
    uint32_t *d_out = C->d_work.as<uint32_t>(), *d_in = d_out + WORDS;
    const int next = (C->rank + 1) % C->world, prev = (C->rank + C->world - 1) % C->world;
    auto word = [](int rank, size_t i) { return (uint32_t)rank * 0x01000193u + (uint32_t)i * 0x9e3779b9u + 7u; };
    std::vector<uint32_t> h(WORDS), got(WORDS, 0);
    for (size_t i = 0; i < WORDS; i++) h[i] = word(C->rank, i);
    PLK_HIP(hipMemcpyAsync(d_out, h.data(), WORDS * sizeof(uint32_t), hipMemcpyHostToDevice, C->stream));
    PLK_HIP(hipMemsetAsync(d_in, 0, WORDS * sizeof(uint32_t), C->stream));
    ShardHeader hd{};
    hd.magic = SHARD_MAGIC; hd.op = 0x74736574u /* "test" */; hd.n = 0x0123456789abcdefull;
    if (C->rank == 0) { memcpy(C->h_hdr, &hd, sizeof hd); PLK_HIP(hipMemcpyAsync(C->d_hdr.p, C->h_hdr, sizeof hd, hipMemcpyHostToDevice, C->stream)); }
    else PLK_HIP(hipMemsetAsync(C->d_hdr.p, 0, sizeof hd, C->stream));
    ncclResult_t e = R->Broadcast(C->d_hdr.p, C->d_hdr.p, sizeof hd, ncclUint8, 0, C->nccl, C->stream);
    if (e != ncclSuccess) return rccl_fail(e, "ncclBroadcast (self-test)");
    e = R->GroupStart();
    if (e != ncclSuccess) return rccl_fail(e, "ncclGroupStart");
    e = R->Send(d_out, WORDS * sizeof(uint32_t), ncclUint8, next, C->nccl, C->stream);
    const ncclResult_t e1 = R->Recv(d_in, WORDS * sizeof(uint32_t), ncclUint8, prev, C->nccl, C->stream);
    const ncclResult_t e2 = R->GroupEnd();
    if (e != ncclSuccess) return rccl_fail(e, "ncclSend (self-test)");
    if (e1 != ncclSuccess) return rccl_fail(e1, "ncclRecv (self-test)");
    if (e2 != ncclSuccess) return rccl_fail(e2, "ncclGroupEnd");
    char *back = static_cast<char *>(C->h_hdr) + sizeof(ShardHeader);
    PLK_HIP(hipMemcpyAsync(back, C->d_hdr.p, sizeof hd, hipMemcpyDeviceToHost, C->stream));
    PLK_TRY(watch_exchange(C, C->stream));
    PLK_HIP(hipMemcpy(got.data(), d_in, WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (memcmp(back, &hd, sizeof hd) != 0) { set_error("plk_comm_selftest: the broadcast header arrived changed"); return PLK_ERR_IO; }
    for (size_t i = 0; i < WORDS; i++)
        if (got[i] != word(prev, i)) { set_error("plk_comm_selftest: the ring step delivered other bytes than rank " + std::to_string(prev) + " sent"); return PLK_ERR_IO; }
    return PLK_OK;
}

// The scatter step of owner-computes mode through the REAL RCCL branch on one GPU (round 6; the TCP tier of the tests synchronises the
// producer before it copies, so it cannot see a missing wait): `iterations` times, a vector of 2^log_n scalars is (re)written on the
// context's stream BEHIND a kernel that parks it for ~0.2 ms — the scalars are still being written when comm_send_work is called, as the
// prover's are —, sent through comm_send_work (event hand-off, header broadcast, grouped ncclSend; the communicator in self-loop: rank 0
// receives its own share inside the same group) and received through comm_recv_work (header check, event hand-off to the consumer's
// stream); the commitment of what arrived must equal the commitment of the vector itself.  *mismatches counts the iterations where it does
// not (a stale vector: with PLK_COMM_TEST_SKIP_PRODUCER_WAIT=1, the negative control, every one).  World 1, RCCL transport, key of >= 2^log_n points.
int32_t plk_comm_scatter_selftest(plk_ctx *ctx, uint32_t log_n, uint32_t iterations, uint32_t *mismatches) {
    Comm *C = ctx ? static_cast<Comm *>(ctx->comm) : nullptr;
    if (!C || C->tcp || C->world != 1 || !mismatches || log_n > 24) { set_error("plk_comm_scatter_selftest: needs an RCCL communicator of ONE rank on this context (plk_comm_init) and log_n <= 24"); return PLK_ERR_ARG; }
    const uint64_t n = 1ull << log_n;
    if (!ctx->srs || ctx->srs_n < n) { set_error("plk_comm_scatter_selftest: the resident key is shorter than 2^log_n"); return PLK_ERR_SRS; }
    if (ctx->msm_enq != ctx->msm_fin) { set_error("plk_comm_scatter_selftest: a commitment is still in flight"); return PLK_ERR_ARG; }
    PLK_HIP(hipSetDevice(ctx->device));
    *mismatches = 0;
    DevBuf v;
    PLK_TRY(v.reserve(n * 32));
    hipStream_t consumer = nullptr;
    PLK_HIP(hipStreamCreateWithFlags(&consumer, hipStreamNonBlocking));
    const bool keep_scatter = C->scatter;
    C->self_loop = true;
    int32_t rc = PLK_OK;
    for (uint32_t it = 0; it < iterations && rc == PLK_OK; it++) {
        hipLaunchKernelGGL(comm_test_stall_kernel, dim3(1), dim3(1), 0, ctx->stream, 20000ull);                 // 0.2 ms at 100 MHz
        hipLaunchKernelGGL(comm_test_fill_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, v.as<uint32_t>(), n, it + 1);
        const void *vec = v.p;
        rc = comm_send_work(ctx, &vec, 1, n, n, false, ctx->stream);
        ShardWork w;
        if (rc == PLK_OK) rc = comm_recv_work(ctx, &w, consumer);
        if (rc == PLK_OK && (w.op != SHARD_COMMIT || w.count != 1 || w.len != n)) { set_error("plk_comm_scatter_selftest: the batch arrived changed"); rc = PLK_ERR_IO; }
        host::HJac got, want;
        if (rc == PLK_OK) rc = msm_enqueue(ctx, static_cast<const Fr *>(w.vec[0]), n, 0, consumer);
        if (rc == PLK_OK) rc = msm_finish(ctx, nullptr, &got);
        if (rc == PLK_OK) rc = msm_enqueue(ctx, v.as<Fr>(), n, 0, ctx->stream);
        if (rc == PLK_OK) rc = msm_finish(ctx, nullptr, &want);
        if (rc == PLK_OK) {
            const host::HAffine a = host::jac_to_affine(got), b = host::jac_to_affine(want);
            if (memcmp(a.x.l, b.x.l, 32) != 0 || memcmp(a.y.l, b.y.l, 32) != 0) ++*mismatches;
        }
    }
    C->self_loop = false;
    C->scatter = keep_scatter;
    (void)hipStreamSynchronize(consumer);
    (void)hipStreamDestroy(consumer);
    (void)hipStreamSynchronize(ctx->stream);
    v.release();
    return rc;
}

// how many ranks RCCL itself counts in this context's communicator (ncclCommCount) — what a multi-GPU bench line prints beside `n_gpus`,
// so that a reader can tell N ranks joined by RCCL from N ranks that fell back to another carrier; 0 without an RCCL communicator
int32_t plk_comm_nccl_count(const plk_ctx *ctx, int32_t *count) {
    if (!ctx || !count) { set_error("plk_comm_nccl_count: bad argument"); return PLK_ERR_ARG; }
    *count = 0;
    const Comm *C = static_cast<const Comm *>(ctx->comm);
    Rccl *R = rccl();
    if (!C || C->tcp || !C->nccl || !R || !R->CommCount) return PLK_OK;
    int k = 0;
    if (R->CommCount(C->nccl, &k) == ncclSuccess) *count = k;
    return PLK_OK;
}

int32_t plk_comm_set_shard(plk_ctx *ctx, uint64_t first_index) {
    if (!ctx || !ctx->comm) { set_error("plk_comm_set_shard: no communicator on this context (plk_comm_init)"); return PLK_ERR_ARG; }
    return plk_set_commit_shard(ctx, first_index, builtin_combine, ctx->comm);
}

int32_t plk_comm_destroy(plk_ctx *ctx) {
    if (!ctx) { set_error("plk_comm_destroy: null ctx"); return PLK_ERR_ARG; }
    comm_release(ctx);
    return PLK_OK;
}

int32_t plk_comm_info(const plk_ctx *ctx, int32_t *rank, int32_t *world, uint64_t *exchanges) {
    if (!ctx) { set_error("plk_comm_info: null ctx"); return PLK_ERR_ARG; }
    const Comm *C = static_cast<const Comm *>(ctx->comm);
    if (rank) *rank = C ? C->rank : 0;
    if (world) *world = C ? C->world : 1;
    if (exchanges) *exchanges = C ? C->gathers : 0;
    return PLK_OK;
}

}  // extern "C"
