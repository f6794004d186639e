// communicator of the multi-GPU commitments (comm.cpp); owned by the context
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
struct plk_ctx;
namespace plk {
void comm_release(plk_ctx *ctx);

// ---- owner-computes ("scatter") mode, round 5: rank 0 alone runs the prover; for every batch of commitments it sends each other
// rank ITS slice of the scalar vectors (N/G x 32 B per vector and link) and gets 96 bytes back.  The transport below is what
// prover.hip's commit_begin (owner) and plk_comm_serve (workers) are built from.
struct ShardWork {
    uint32_t op = 0;                 // SHARD_COMMIT / SHARD_STOP
    uint32_t count = 0;              // vectors of the batch (1..8)
    uint32_t lagrange = 0;           // commit against the Lagrange-form key
    uint64_t n = 0, slice = 0;       // length of the owner's vectors, points per rank
    uint64_t len = 0;                // this rank's share: clamp(n - rank * slice, 0, slice)
    const void *vec[8] = {nullptr};  // worker: the received slices (device), valid until the next comm_recv_work
};
constexpr uint32_t SHARD_COMMIT = 1, SHARD_STOP = 2;
bool comm_scatter_owner(const plk_ctx *ctx);         // scatter mode, this is rank 0 of a communicator of more than one rank
bool comm_scatter_worker(const plk_ctx *ctx);        // scatter mode, rank > 0
// owner: header + every worker's slice of vecs[0..count) (device pointers, n elements each), ordered after `producer`
int32_t comm_send_work(plk_ctx *ctx, const void *const *vecs, uint32_t count, uint64_t n, uint64_t slice, bool lagrange, hipStream_t producer);
// worker: blocks (no deadline: the owner may be busy or idle for any time) until the owner sends work or stops
int32_t comm_recv_work(plk_ctx *ctx, ShardWork *w, hipStream_t consumer);
int32_t comm_send_stop(plk_ctx *ctx);
}
