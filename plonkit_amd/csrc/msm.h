#pragma once
#include "ctx.h"
#include "hostmath.h"
namespace plk {
// enqueue the whole Pippenger pipeline on `stream`; results (window sums) are copied to ctx->pinned
int32_t msm_enqueue(plk_ctx *ctx, const Fr *scalars_dev, uint64_t n, uint64_t base_offset, hipStream_t stream);
// synchronise and fold the window sums on the host
int32_t msm_finish(plk_ctx *ctx, hipStream_t stream, host::HJac *out);
// up to 8 scalar vectors of the same length against the same bases: one pass of every kernel
int32_t msm_enqueue_batch(plk_ctx *ctx, const Fr *const *scalars_dev, uint32_t batch, uint64_t n, uint64_t base_offset, hipStream_t stream);
int32_t msm_finish_batch(plk_ctx *ctx, hipStream_t stream, host::HJac *out);
// 0, or the piece size in which a commitment of `terms` terms against the resident SRS should be run with several pieces in
// flight (2^20 from 2^23 terms on, when the SRS has its table of shifted copies: without them short pieces need more windows)
uint64_t msm_pipelined_piece(const plk_ctx *ctx, uint64_t terms);
int32_t ensure_pinned(plk_ctx *ctx, size_t bytes);
}  // namespace plk
